"""WindowedCache (reference: awq/modules/fused/cache.py:4-79): fp16 K / V stores of shape
[batch, max_seq_len, n_kv_heads, head_dim] that roll once max_seq_len is exceeded.  Device memory
plumbing only; the kernels write rows through `ops.rope_kv_append` and read them through
`ops.decode_attention`."""
import torch


class WindowedCache:
    def __init__(self, cache_batch_size, n_heads, n_kv_heads, head_dim, max_seq_len, device):
        size = (cache_batch_size, max_seq_len, n_kv_heads if n_kv_heads != 0 else n_heads, head_dim)
        self.v = torch.zeros(size, device=device, dtype=torch.float16)
        self.k = torch.zeros(size, device=device, dtype=torch.float16)
        self.max_seq_len = max_seq_len

    def get_kv(self, batch_size, start_pos, seqlen):
        return self.v[:batch_size, : start_pos + seqlen], self.k[:batch_size, : start_pos + seqlen]

    def update_kv(self, values_store, keys_store, batch_size, start_pos, seqlen):
        self.v[:batch_size, start_pos: start_pos + seqlen, :, :] = values_store
        self.k[:batch_size, start_pos: start_pos + seqlen, :, :] = keys_store

    def roll_kv_n_steps(self, start_pos, n=100):
        """Drop the n oldest positions (sequence axis = dim 1 of this layout) and zero the freed tail."""
        n = min(n, self.max_seq_len)
        self.v = torch.roll(self.v, shifts=-n, dims=1)
        self.k = torch.roll(self.k, shifts=-n, dims=1)
        self.v[:, -n:, :, :] = 0
        self.k[:, -n:, :, :] = 0
        return start_pos - n

    def to(self, device):
        self.k = self.k.to(device)
        self.v = self.v.to(device)

    def increase_batch_size(self, to_bsz):
        self.v = torch.zeros(to_bsz, *self.v.shape[1:], dtype=self.v.dtype, device=self.v.device)
        self.k = torch.zeros(to_bsz, *self.k.shape[1:], dtype=self.k.dtype, device=self.k.device)

    def decrease_batch_size(self, to_bsz):
        self.v = self.v[:to_bsz].contiguous()
        self.k = self.k[:to_bsz].contiguous()
