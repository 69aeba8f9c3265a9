"""GraphedDecoder: one decode step = one hipGraph replay, for a fused model built by autoawq_amd.fuser (round 4).

The reference decodes by calling the fused model once per token from `generate` (awq/modules/fused/model.py:60-109,
examples/benchmark.py:55-87): every step re-issues ~160 launches from Python.  On MI355X a 7B decode step is ~1.2 ms of GPU
time in ~160 launches of 3-10 us each; launching them from the host costs more than running them.  So a step is captured
ONCE and replayed:

  * the positions live in two device words (start position, cache length) that the captured step itself advances, so one
    graph serves every token (QuantAttentionFused.use_device_positions);
  * a launch inside a graph has a FIXED grid, but the best attention launch depends on the context length: a 32-way split of
    the cache with a merge kernel behind it is right at 4k tokens and costs 14 us per layer at 64.  The decoder therefore keeps
    one graph per LENGTH BUCKET (256, 1024, 4096, ... cache rows): the host knows the position (it drives the loop), picks the
    bucket, and the attention launch inside that graph is sized for the bucket's bound (one split and no merge launch up to
    256 rows); within a bucket the kernel deals the rows that exist to its splits from the device-side length;
  * the token ids and the logits are static buffers (`step` copies the ids in, returns the logits tensor of the graph);
  * a step that would run past the cache window forgets the oldest StepPlan.ROLL positions first (the reference's policy,
    awq/utils/fused_utils.py:14-25), eagerly, between two replays.

    dec = GraphedDecoder(fused_lm, batch=1)
    logits = dec.prefill(prompt_ids)          # eager, any length
    for _ in range(n):
        logits = dec.step(next_ids)           # [batch, 1, vocab]; valid until the next step
"""
import torch

from .model import StepPlan


class GraphedDecoder:
    BUCKETS = (256, 1024, 4096, 16384, 65536, 262144)

    def __init__(self, lm, batch=1, buckets=None):
        self.lm = lm
        self.blocks = list(lm.model.blocks)
        self.device = self.blocks[0].device if hasattr(self.blocks[0], "device") else next(lm.parameters()).device
        self.device = torch.device(self.device)
        if self.device.type != "cuda":
            raise RuntimeError("GraphedDecoder needs the model on a HIP device")
        self.batch = int(batch)
        # A captured step bakes every HOST-side length into the graph.  Only the native decode path of QuantAttentionFused (head_dim
        # 128 with 1, 2, 4 or 8 query heads per KV head: csrc/decoder.hip) reads its positions from the device words; other head
        # shapes slice the cache with the host's start_pos and would attend over the capture-time row count on every replay
        # (ADVICE r04): refuse them here instead of returning silently wrong logits.
        for i, b in enumerate(self.blocks):
            a = b.attn
            if a.head_dim != 128 or a.n_kv_groups not in (1, 2, 4, 8):
                raise NotImplementedError(f"GraphedDecoder: block {i} has head_dim {a.head_dim} and {a.n_kv_groups} query heads per KV head; "
                                          "only the native decode attention (head_dim 128; 1, 2, 4 or 8 heads per KV head) takes its "
                                          "positions from the device -- decode this model eagerly")
        self.max_seq_len = min(b.attn.max_seq_len for b in self.blocks)
        bounds = sorted(set(min(int(b), self.max_seq_len) for b in (buckets or self.BUCKETS)))
        if bounds[-1] < self.max_seq_len:
            bounds.append(self.max_seq_len)
        self.bounds = bounds
        with torch.inference_mode(False):  # persistent buffers written in place from any context: never inference tensors
            self.pos = torch.zeros(1, dtype=torch.int32, device=self.device)   # start position of the step
            self.len = torch.ones(1, dtype=torch.int32, device=self.device)    # cache rows after its append
            self.tok = torch.zeros((self.batch, 1), dtype=torch.int64, device=self.device)
        self.stream = torch.cuda.Stream(device=self.device)
        self.graphs = {}  # bucket bound -> (graph, logits)
        self.position = 0
        for b in self.blocks:
            b.attn._resize_cache(self.batch)

    # ---- bookkeeping ------------------------------------------------------------------------------------------------
    def _sync_host_state(self):
        for b in self.blocks:
            b.attn.start_pos = self.position
        self.lm.model.last_forward_num_tokens = self.position

    def seek(self, position):
        """Declare that `position` cache rows are valid (benchmarks and tests that fill the caches themselves)."""
        if not 0 <= position <= self.max_seq_len:  # (== max_seq_len: a full window -- the next step rolls it first)
            raise ValueError(f"position {position} outside the cache window {self.max_seq_len}")
        self.position = int(position)
        self.pos.fill_(self.position)
        self.len.fill_(self.position + 1)
        self._sync_host_state()

    def bucket(self, rows):
        for b in self.bounds:
            if rows <= b:
                return b
        return self.bounds[-1]

    # ---- prefill: eager (any number of tokens; the reference's own path) ----------------------------------------------
    @torch.inference_mode()
    def prefill(self, input_ids):
        for b in self.blocks:
            b.attn.use_device_positions(None, None)
            b.attn.start_pos = 0
        self.lm.model.last_forward_num_tokens = 0
        logits = self.lm(input_ids.to(self.device))
        self.seek(self.blocks[0].attn.start_pos)
        return logits

    # ---- one token ------------------------------------------------------------------------------------------------------
    @torch.inference_mode()
    def _capture(self, bound):
        for b in self.blocks:
            b.attn.use_device_positions(self.pos, self.len)
            b.attn.decode_len_bound = bound
        keep = (self.pos.clone(), self.len.clone())
        caller = torch.cuda.current_stream(self.device)  # (taken BEFORE entering the decoder's stream: the ids were copied on it)
        with torch.cuda.stream(self.stream):
            self.stream.wait_stream(caller)
            # (the warm-up step runs the kernels for real -- it appends a row at the current position, which the first replay
            #  writes again with the real token -- and builds every lazily cached buffer OUTSIDE the capture)
            self.lm(self.tok)
            self._sync_host_state()
            self.stream.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=self.stream):
                logits = self.lm(self.tok)
                self.pos.add_(1)
                self.len.add_(1)
            self.stream.synchronize()
            self.pos.copy_(keep[0])
            self.len.copy_(keep[1])
        caller.wait_stream(self.stream)
        self._sync_host_state()
        self.graphs[bound] = (graph, logits)
        return self.graphs[bound]

    def _advance(self):
        """What `step` and `replay` share: roll a full window, pick (or capture) the bucket's graph, replay it on the decoder's stream
        between the caller's work, advance the position on the host as the graph did on the device."""
        if self.position + 1 > self.max_seq_len:  # past the window: forget the oldest positions (eagerly)
            live = [b.attn.cache.roll_kv_n_steps(self.position, n=StepPlan.ROLL) for b in self.blocks][0]
            self.seek(live)
        bound = self.bucket(self.position + 1)
        graph, logits = self.graphs.get(bound) or self._capture(bound)
        caller = torch.cuda.current_stream(self.device)
        self.stream.wait_stream(caller)   # the ids were copied on the caller's stream
        with torch.cuda.stream(self.stream):
            graph.replay()
        caller.wait_stream(self.stream)   # ... and the caller reads the logits
        self.position += 1
        self._sync_host_state()
        return logits

    @torch.inference_mode()
    def step(self, token_ids):
        """token_ids [batch, 1] (or [batch]): the tokens at the current position.  Returns the logits [batch, 1, vocab] -- the
        graph's static output buffer, overwritten by the next step."""
        self.tok.copy_(token_ids.reshape(self.batch, 1))
        return self._advance()

    @torch.inference_mode()
    def replay(self):
        """The current bucket's graph once more on the decoder's stream with the ids already in place (benchmarks: no host
        copies in the timed region).  Same roll / stream / host bookkeeping as `step`."""
        return self._advance()
