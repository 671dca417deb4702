"""Continuous batching on the paged KV cache (the serving half).

The reference serves one request at a time (`app.py:1128`: Gradio queue, concurrency 1; global `model` / `conv`). With a paged
cache and packed sequences a second request does not have to wait for the first one's 1024 decode steps: every step of this
engine decodes ONE token for every active sequence in a single `vt_llama_forward` call (one row per sequence, weight-streaming
GEMMs + the fused decode attention), and requests join / leave between steps.

  eng = ServingEngine(model, max_batch=16, kv_pages=512)
  a = eng.submit(ids_a, images=[img], regions=[box], max_new_tokens=256)
  b = eng.submit(ids_b, max_new_tokens=64)           # may be submitted while `a` is already decoding
  for rid, token in eng.step(): ...                  # or: outputs = eng.run()   -> {rid: LongTensor of new tokens}

Admission runs the request's multimodal prefill on its own (towers -> splice -> decoder prefill, exactly what `generate` does,
including the visual-feature cache), so a request's tokens are bit-identical to running it alone with greedy decoding as long
as the decode batch stays on one kernel path (<= 16 rows: the folded-norm path; the engine never mixes a sequence between the
<= 16 and > 16 row paths within one request unless max_batch > 16).

Failures are isolated per request: a request whose own multimodal encoding or prefill fails (bad image shape, bad region, a prompt
beyond the rotary table) is moved to `failed` with its exception and leaves the queue -- the requests behind it are served. Only
errors that are not tied to one request (a HIP runtime error) propagate out of step(), with every page this call took given back
and the candidates re-queued. A pool whose pages are held by someone else (a `past_key_values` the caller keeps) makes the waiting
requests WAIT: step() returns [] and run() raises "no progress" instead of spinning when nothing is active and nothing can be admitted. `kv_pages` given to the constructor is an UPPER BOUND of the pool: the
engine never rebuilds a larger one behind the caller's back; requests that do not fit next to the running ones wait, and one that
cannot fit even an empty pool of that size fails instead of blocking the queue forever.
"""
from __future__ import annotations

import collections
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch

from . import ops
from .engine import PagedKVCache, SequenceState, llama_forward, pair_lo


@dataclass
class _Request:
    rid: int
    input_ids: torch.Tensor
    images: Optional[list]
    regions: Optional[list]
    max_new_tokens: int
    eos: frozenset
    seq: SequenceState = field(default_factory=SequenceState)
    tokens: List[int] = field(default_factory=list)
    last: Optional[torch.Tensor] = None      # device int32 [1]: the token to feed next
    done: bool = False
    flat: Optional[torch.Tensor] = None      # spliced prompt rows [rows, H] (kept while the request waits for pages)
    flat_lo: Optional[torch.Tensor] = None   # their low halves in the precise levels 2 / 3 (the embeddings travel as an operand pair)
    need: int = 0                            # KV pages for prompt + max_new_tokens
    error: Optional[BaseException] = None    # set when the request failed on its own (it is in ServingEngine.failed then)


class ServingEngine:
    def __init__(self, model, max_batch: int = 16, kv_pages: Optional[int] = None, do_sample: bool = False,
                 temperature: float = 1.0, top_p: float = 1.0, seed: int = 0, batch_prefill: bool = False):
        self.model = model
        self.max_batch = int(max_batch)
        self.do_sample, self.temperature, self.top_p, self.seed = bool(do_sample), float(temperature), float(top_p), int(seed)
        # batch_prefill: requests admitted in the same step share ONE packed decoder prefill (larger GEMMs: higher throughput);
        # their logits then differ from a solo run by bf16 noise (other GEMM tiles), so it is off where bit-identity matters
        self.batch_prefill = bool(batch_prefill)
        self.waiting: "collections.deque[_Request]" = collections.deque()
        self.active: List[_Request] = []
        self.finished: Dict[int, _Request] = {}
        self.failed: Dict[int, _Request] = {}            # requests that failed on their own (`.error` holds the exception)
        self._next_id = 0
        self._step = 0
        self.kv_cap: Optional[int] = None                # upper bound of the pool (pages) when the caller fixed one
        if kv_pages is not None:
            model.reset_prefix_cache()
            model.kv = PagedKVCache(model.get_model().llama, int(kv_pages))
            self.kv_cap = int(kv_pages)
        from .prefix_cache import VisualFeatureCache
        self._vis_cache = VisualFeatureCache(int(getattr(model.config, "vis_cache_entries", 16)))

    # ---- queue ----------------------------------------------------------------------------------------------------------
    def submit(self, input_ids: torch.Tensor, images=None, regions=None, max_new_tokens: int = 256,
               eos_token_id=None) -> int:
        ids = input_ids if input_ids.dim() == 2 else input_ids.unsqueeze(0)
        if ids.shape[0] != 1:
            raise ValueError("submit() takes one sequence per request")
        eos = self.model.config.eos_token_id if eos_token_id is None else eos_token_id
        eos_set = frozenset(eos) if isinstance(eos, (list, tuple, set, frozenset)) else (frozenset([eos]) if eos is not None else frozenset())
        r = _Request(self._next_id, ids.to(self.model.device), images, regions, int(max_new_tokens), eos_set)
        self._next_id += 1
        self.waiting.append(r)
        return r.rid

    def pending(self) -> int:
        return len(self.waiting) + len(self.active)

    def cancel(self, rid: int) -> bool:
        """Drop a request that is still waiting (True) -- a running or finished one is left alone (False)."""
        for r in self.waiting:
            if r.rid == rid:
                self.waiting.remove(r)
                r.flat = None
                return True
        return False

    def _fail(self, r: _Request, exc: BaseException) -> None:
        """The request failed on its own: it leaves the engine with its exception; nothing else is disturbed."""
        if r.seq.pages:
            self.model.kv.release(r.seq.pages)
        r.seq.pages, r.seq.length, r.last, r.flat = [], 0, None, None
        r.error, r.done = exc, True
        self.failed[r.rid] = r

    @staticmethod
    def _is_device_error(exc: BaseException) -> bool:
        """Errors that are not one request's fault: the HIP runtime (status -2 of the C ABI), the allocator."""
        return isinstance(exc, torch.cuda.OutOfMemoryError) or "status -2" in str(exc) or "HIP error" in str(exc)

    # ---- one scheduling step ---------------------------------------------------------------------------------------------
    def _pick(self, logits: torch.Tensor) -> torch.Tensor:
        if self.do_sample:
            return ops.sample_top_p(logits, self.temperature, self.top_p, self.seed, self._step)
        return ops.argmax(logits)

    def _embed(self, r: _Request) -> torch.Tensor:
        m = self.model
        (_, _, _, _, embeds, _) = m.prepare_inputs_labels_for_multimodal(r.input_ids, None, None, None, None, r.images, r.regions,
                                                                         feature_cache=self._vis_cache if r.images is not None else None)
        if embeds is None:
            return m.get_model().embed_tokens(r.input_ids)[0]
        mask = torch.tensor(m._last_splice[0][0], dtype=torch.bool, device=embeds.device)
        lo = pair_lo(embeds)                  # precise levels 2 / 3: the spliced embeddings are an operand pair -- the low half travels explicitly
        r.flat_lo = None if lo is None else lo[0][mask]          # (a tensor attribute would not survive the indexing below: ADVICE r5)
        return embeds[0][mask]

    def _admit(self, reqs: List[_Request]) -> List[_Request]:
        """Multimodal prefill of the requests that fit: towers + splice per request (once: the rows and the page need stay on the
        request while it waits), decoder prefill per request (default) or one packed pass for all of them (batch_prefill). Every
        admitted request RESERVES its worst case up front -- pages for prompt + max_new_tokens are allocated here, so a sequence
        that is already decoding can never find the pool empty because a later admission took its pages. When nothing of this
        engine is live the pool is (re)built for the candidate batch, never beyond the constructor's `kv_pages`; if it cannot be
        grown (out of memory, pages held by someone else) the prefix of `reqs` that fits the pool there is gets admitted and the rest
        waits, as requests that do not fit next to running ones always do. Returns the requests that stay queued, in order.
        A request that fails on its own is moved to `failed` (module docstring); a device-level error propagates with the engine left
        as it was: pages taken by this call released, the surviving candidates back at the head of the queue."""
        m = self.model
        llama = m.get_model().llama
        live: List[_Request] = []                           # candidates that are still in the game
        for r in reqs:
            if r.flat is None:
                try:
                    r.flat = self._embed(r)
                    r.need = (r.flat.shape[0] + r.max_new_tokens + 63) // 64 + 1
                except Exception as e:  # noqa: BLE001 -- classified below
                    if self._is_device_error(e):
                        for q in reversed([x for x in reqs if x.error is None]):
                            self.waiting.appendleft(q)
                        raise
                    self._fail(r, e)                        # bad image shape / region / ids: this request's own problem
                    continue
            if self.kv_cap is not None and r.need > self.kv_cap:      # larger than the pool may ever be: no point in waiting
                self._fail(r, RuntimeError(f"request {r.rid} needs {r.need} KV pages, the pool is capped at {self.kv_cap}"))
                continue
            live.append(r)
        admitted: List[_Request] = []
        try:
            if not self.active and live:
                # idle: size the pool for every candidate (the first len(reqs) <= max_batch requests decode together), within the cap
                total = sum(r.need for r in live)
                want = total if self.kv_cap is None else min(total, self.kv_cap)
                if m.kv is None or len(m.kv.free) < want:
                    try:
                        m._ensure_kv(want)
                    except (RuntimeError, torch.cuda.OutOfMemoryError) as e:
                        if m.kv is None or (self._is_device_error(e) and not isinstance(e, torch.cuda.OutOfMemoryError)):
                            raise                           # no pool at all, or a HIP runtime error: nothing can be scheduled
                        # the pool could not grow: go on with the one that is there (the prefix that fits is admitted below)
            for r in live:
                if len(m.kv.free) < r.need:
                    if not admitted and not self.active and len(m.kv.free) == m.kv.num_pages:
                        # nothing is running, every page of the pool is free, the pool is as large as it could be made (the cap, or
                        # what the allocator gave), and the head request still does not fit: it never will -- fail it instead of
                        # blocking everything behind it
                        self._fail(r, RuntimeError(f"request {r.rid} needs {r.need} KV pages, the pool holds {m.kv.num_pages}"
                                                   + (f" (capped at {self.kv_cap})" if self.kv_cap is not None else "")))
                        continue
                    break                                   # wait for running requests to retire (or for an idle engine to grow the pool)
                r.seq.pages = m.kv.alloc(r.need)
                admitted.append(r)
        except Exception:
            for r in admitted:
                m.kv.release(r.seq.pages)
                r.seq.pages = []
            for r in reversed(live):
                if r.error is None:
                    self.waiting.appendleft(r)
            raise
        rest = [r for r in live if r.error is None and r not in admitted]
        ok: List[_Request] = []

        def prefill_one(r):
            try:
                r.last = self._pick(llama_forward(llama, m.kv, [r.seq], r.flat, [r.flat.shape[0]], embeds_lo=getattr(r, "flat_lo", None)))
                ok.append(r)
            except Exception as e:  # noqa: BLE001
                if self._is_device_error(e):
                    raise
                self._fail(r, e)                            # e.g. a prompt beyond the rotary table: this request's own problem

        try:
            if self.batch_prefill and len(admitted) > 1:
                try:
                    flats = [r.flat for r in admitted]
                    los = [getattr(r, "flat_lo", None) for r in admitted]
                    lo_cat = None
                    if any(l is not None for l in los):      # (requests without visual rows have an exact 16-bit embedding: low half zero)
                        lo_cat = torch.cat([l if l is not None else torch.zeros_like(f) for l, f in zip(los, flats)], 0)
                    logits = llama_forward(llama, m.kv, [r.seq for r in admitted], torch.cat(flats, 0), [f.shape[0] for f in flats], embeds_lo=lo_cat)
                    nxt = self._pick(logits)
                    for i, r in enumerate(admitted):
                        r.last = nxt[i:i + 1]
                    ok = list(admitted)
                except Exception as e:  # noqa: BLE001
                    if self._is_device_error(e):
                        raise
                    for r in admitted:                      # find the culprit: every request on its own (a solo prefill is always legal)
                        r.seq.length = 0
                        prefill_one(r)
            else:
                for r in admitted:
                    prefill_one(r)
        except Exception:                                   # device-level: pages back, every surviving candidate queued again
            for r in admitted:
                if r.error is None:
                    m.kv.release(r.seq.pages)
                    r.seq.pages, r.seq.length, r.last = [], 0, None
            for r in reversed(live):
                if r.error is None:
                    self.waiting.appendleft(r)
            raise
        for r in ok:
            r.flat = r.flat_lo = None                       # the rows are in the cache now
        self.active += ok
        return rest

    def _retire(self, r: _Request) -> None:
        r.done = True
        self.model.kv.release(r.seq.pages)
        r.seq.pages = []
        self.finished[r.rid] = r

    def step(self) -> List[Tuple[int, int]]:
        """Admit waiting requests while there is room, emit one token for every active request. Returns [(request id, token)]."""
        cand: List[_Request] = []
        while self.waiting and len(self.active) + len(cand) < self.max_batch:
            cand.append(self.waiting.popleft())
        if cand:
            for r in reversed(self._admit(cand)):          # what did not fit keeps its place at the head of the queue
                self.waiting.appendleft(r)
        if not self.active:
            return []
        # tokens chosen at the end of the previous step (or by the prefill) become visible now: ONE read-back per step
        toks = torch.cat([r.last for r in self.active]).tolist()
        out: List[Tuple[int, int]] = []
        still: List[_Request] = []
        for r, t in zip(self.active, toks):
            r.tokens.append(int(t))
            out.append((r.rid, int(t)))
            if int(t) in r.eos or len(r.tokens) >= r.max_new_tokens:
                self._retire(r)
            else:
                still.append(r)
        self.active = still
        if self.active:
            m = self.model
            llama = m.get_model().llama
            ids = torch.cat([r.last for r in self.active]).to(torch.int32)
            plan = torch.stack([torch.zeros_like(ids), ids], dim=1).contiguous()
            x = ops.embed_splice(llama.embed, None, None, plan)
            logits = llama_forward(llama, m.kv, [r.seq for r in self.active], x, [1] * len(self.active))
            nxt = self._pick(logits)
            for i, r in enumerate(self.active):
                r.last = nxt[i:i + 1]
        self._step += 1
        return out

    def run(self, on_token=None) -> Dict[int, torch.Tensor]:
        """Drive step() until every submitted request has finished; returns {request id: LongTensor of generated tokens}."""
        while self.pending():
            before = (len(self.waiting), len(self.failed), len(self.finished))
            emitted = self.step()
            for rid, t in emitted:
                if on_token is not None:
                    on_token(rid, t)
            if not emitted and not self.active and (len(self.waiting), len(self.failed), len(self.finished)) == before:
                # nothing admitted, nothing failed, nothing running: the KV pages the waiting requests need are held OUTSIDE this
                # engine (e.g. a past_key_values the caller keeps alive) and no later step can change that -- do not spin
                free = len(self.model.kv.free) if self.model.kv is not None else 0
                raise RuntimeError(f"ServingEngine.run: no progress -- {len(self.waiting)} request(s) waiting, none active, {free} KV pages free "
                                   f"(the head request needs {self.waiting[0].need}); pages are held outside the engine")
        return {rid: torch.tensor(r.tokens, dtype=torch.long) for rid, r in sorted(self.finished.items())}

    def errors(self) -> Dict[int, BaseException]:
        """{request id: exception} of the requests that failed on their own."""
        return {rid: r.error for rid, r in sorted(self.failed.items())}
