"""KV store of the fused attention block for MI355X.

Role of `WindowedCache` (reference: awq/modules/fused/cache.py:4-79) with the same method surface --
`get_kv`, `update_kv`, `roll_kv_n_steps`, `to`, `increase_batch_size`, `decrease_batch_size`, attributes
`k`, `v`, `max_seq_len` -- but its own design:

  * ONE allocation `kv[2, batch, max_seq_len, kv_heads, head_dim]` fp16; `k` / `v` are views of it (the
    decode kernels take two base pointers, the sequence axis is dim 1 of each view: a cache row of one
    (token, head) is 256 contiguous bytes, what `awq_decode_attention` reads per lane group);
  * dropping the n oldest positions moves only the LIVE rows (`start_pos - n` of them) forward with one
    strided device copy through a scratch of exactly that size.  The reference rolls and zero-fills the
    whole buffer (two full-size copies per layer); rows at or past the new length are never read
    (`awq_decode_attention` stops at the length), so nothing is cleared;
  * batch-size changes keep the prefix of sequences that survive instead of discarding everything.
"""
import torch


class WindowedCache:
    def __init__(self, cache_batch_size, n_heads, n_kv_heads, head_dim, max_seq_len, device):
        heads = n_kv_heads if n_kv_heads != 0 else n_heads
        self.max_seq_len = int(max_seq_len)
        self._alloc(int(cache_batch_size), heads, int(head_dim), device)

    def _alloc(self, batch, heads, head_dim, device, keep=None):
        with torch.inference_mode(False):  # a persistent buffer written in place from any context: never an inference tensor
            kv = torch.zeros((2, batch, self.max_seq_len, heads, head_dim), dtype=torch.float16, device=device)
            if keep is not None:
                n = min(batch, keep.shape[1])
                kv[:, :n] = keep[:, :n].to(device)
        self.kv = kv
        self.k, self.v = kv[0], kv[1]

    # ---- reference surface -------------------------------------------------------------------
    def get_kv(self, batch_size, start_pos, seqlen):
        end = start_pos + seqlen
        return self.v[:batch_size, :end], self.k[:batch_size, :end]

    def update_kv(self, values_store, keys_store, batch_size, start_pos, seqlen):
        rows = slice(start_pos, start_pos + seqlen)
        self.k[:batch_size, rows].copy_(keys_store)
        self.v[:batch_size, rows].copy_(values_store)

    def roll_kv_n_steps(self, start_pos, n=100):
        """Forget the n oldest positions; returns the new start position (reference: cache.py:47-62)."""
        n = max(0, min(int(n), int(start_pos), self.max_seq_len))
        live = int(start_pos) - n
        if n and live > 0:
            moved = self.kv[:, :, n:n + live].clone()  # source and destination overlap: go through a scratch
            self.kv[:, :, :live].copy_(moved)
        return live

    def to(self, device):
        self._alloc(self.kv.shape[1], self.kv.shape[3], self.kv.shape[4], device, keep=self.kv)

    def increase_batch_size(self, to_bsz):
        self._alloc(int(to_bsz), self.kv.shape[3], self.kv.shape[4], self.kv.device, keep=self.kv)

    def decrease_batch_size(self, to_bsz):
        self._alloc(int(to_bsz), self.kv.shape[3], self.kv.shape[4], self.kv.device, keep=self.kv)
