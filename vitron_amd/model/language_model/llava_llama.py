"""LlavaLlamaForCausalLM on HIP kernels -- mirror of the reference's vitron/model/language_model/llava_llama.py.

Same class names (LlavaConfig, LlavaLlamaModel, LlavaLlamaForCausalLM) and the same forward / generate /
prepare_inputs_for_generation signatures (llava_llama.py:57-114), so app.py:562-571 and
inference_image.py:53-61 call it unchanged. The decoder itself (transformers-4.31 LlamaForCausalLM in the
reference) is vt_llama_forward: one C call per pass over a paged KV cache; `generate` is a greedy / top-p loop on
top of it (the reference gets its loop from GenerationMixin).
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import List, Optional

import torch

from ... import _lib, ops, synth
from ...engine import DecodeState, PackedLlama, PagedKVCache, SequenceState, llama_forward, pair_lo
from ..llava_arch import LlavaMetaForCausalLM, LlavaMetaModel
from ..multimodal_encoder.builder import build_image_tower, build_video_tower
from ..multimodal_projector.builder import build_vision_projector
from ..region_extractor.builder import build_region_extractor


def _resolve_dtype(x, strict=True):
    """torch dtype for a `torch_dtype` entry ('float16' / 'bfloat16' strings as in a HF config.json, torch dtypes,
    'fp16' / 'bf16'); None stays None (= the package default). strict=False (a value READ FROM A CHECKPOINT's config.json): anything
    that is not one of the two operand formats -- "float32", unknown strings -- is treated as None instead of raising: the reference
    ignores the stored value and forces fp16 (builder.py:47), so a checkpoint saved with "torch_dtype": "float32" must still load."""
    if x is None:
        return None
    if isinstance(x, str):
        x = {"float16": torch.float16, "half": torch.float16, "fp16": torch.float16,
             "bfloat16": torch.bfloat16, "bf16": torch.bfloat16}.get(x.replace("torch.", ""), x)
    if not strict and x not in (torch.float16, torch.bfloat16):
        return None
    _lib.operand_of(x)
    return x


class LlavaConfig(SimpleNamespace):
    """LlamaConfig fields + the mm_* fields the reference reads (llava_arch.py:33-40,363,379)."""
    model_type = "llava"

    def __init__(self, **kw):
        base = dict(hidden_size=4096, intermediate_size=11008, num_hidden_layers=32, num_attention_heads=32,
                    vocab_size=32000, rms_norm_eps=1e-5, rope_theta=10000.0, max_position_embeddings=4096,
                    bos_token_id=1, eos_token_id=2, pad_token_id=0, mm_image_tower=None, mm_video_tower=None,
                    mm_projector_type="mlp2x_gelu", mm_vision_select_layer=-2, mm_vision_select_feature="patch",
                    mm_hidden_size=1024, tokenizer_model_max_length=None, tokenizer_padding_side="right",
                    pretraining_tp=1, use_cache=True)
        base.update(kw)
        super().__init__(**base)

    def to_dict(self):
        return dict(self.__dict__)


class CausalLMOutputWithPast(SimpleNamespace):
    def __getitem__(self, i):
        return [v for v in (self.loss, self.logits, self.past_key_values) if v is not None][i]


class PagedPast:
    """past_key_values of this implementation: per-sequence page lists in the model's paged KV pool."""

    def __init__(self, kv: PagedKVCache, seqs: List[SequenceState], padded_len: int):
        self.kv, self.seqs, self.padded_len = kv, seqs, padded_len

    def seq_length(self) -> int:  # what the reference reads as past_key_values[-1][-1].shape[-2] (llava_arch.py:198)
        return self.padded_len

    def release(self):
        for s in self.seqs:
            self.kv.release(s.pages)
            s.pages = []


class LlavaLlamaModel(LlavaMetaModel):
    """reference llava_llama.py:33-37 + LlavaMetaModel.__init__ (llava_arch.py:30-40)."""

    def __init__(self, config: LlavaConfig):
        self.config = config
        self.llama: Optional[PackedLlama] = None
        if getattr(config, "mm_image_tower", None) is not None:
            self.image_tower = build_image_tower(config, delay_load=True)
        if getattr(config, "mm_video_tower", None) is not None:
            self.video_tower = build_video_tower(config, delay_load=True)
        if getattr(config, "mm_image_tower", None) is not None or getattr(config, "mm_video_tower", None) is not None:
            self.mm_projector = build_vision_projector(config)
            self.region_extractor = build_region_extractor(config)

    @property
    def embed_tokens_weight(self) -> torch.Tensor:
        return self.llama.embed

    def embed_tokens(self, ids: torch.Tensor) -> torch.Tensor:
        """nn.Embedding forward. Like the reference's, an id outside the table is an error (IndexError), checked on a host
        copy of the ids (prompts only: the decode loop feeds tokens through vt_decode_feed, which cannot leave the table)."""
        host = ids.detach().reshape(-1).cpu()
        if host.numel() and (int(host.min()) < 0 or int(host.max()) >= self.llama.embed_rows):
            bad = int(host.min()) if int(host.min()) < 0 else int(host.max())
            raise IndexError(f"embed_tokens: token id {bad} outside the embedding table ({self.llama.embed_rows} rows)")
        flat = ids.reshape(-1).to(device=self.llama.device, dtype=torch.int32)
        plan = torch.stack([torch.zeros_like(flat), flat], dim=1).contiguous()
        return ops.embed_splice(self.llama.embed, None, None, plan).view(*ids.shape, -1)


class LlavaLlamaForCausalLM(LlavaMetaForCausalLM):
    config_class = LlavaConfig

    def __init__(self, config: LlavaConfig):
        self.config = config
        self.model = LlavaLlamaModel(config)
        self.vocab_size = config.vocab_size
        self._device = torch.device("cpu")
        self._llama_sd = None
        # operand dtype of the packed weights and activations: bf16 (package default, BASELINE's dtype) or fp16 (the reference's
        # inference dtype, builder.py:47); set by config.torch_dtype / .to(dtype=...) / .half() before the weights are packed
        # (a config.json value is advisory: unsupported entries fall back to the default; explicit .to(dtype=) / torch_dtype= arguments are validated)
        self._dtype = _resolve_dtype(getattr(config, "torch_dtype", None), strict=False)
        self.kv: Optional[PagedKVCache] = None
        self.kv_pages = getattr(config, "kv_pages", None)
        # multi-turn reuse (vitron_amd/prefix_cache.py): generate() keeps the last conversation's KV pages and the encoded
        # images; config.kv_prefix_reuse / config.vis_cache_entries switch them off (benchmarks do)
        self._prefix = None
        self._vis_cache = None
        self.last_generate_stats = {}

    # ---- module-like plumbing -----------------------------------------------------------------------------------
    def get_model(self):
        return self.model

    @property
    def device(self):
        return self._device

    @property
    def dtype(self):
        if self.model.llama is not None:
            return self.model.llama.dtype
        return self._dtype or _lib.torch_dtype()

    def half(self):
        return self.to(dtype=torch.float16)

    def bfloat16(self):
        return self.to(dtype=torch.bfloat16)

    def eval(self):
        return self

    def requires_grad_(self, flag=False):
        return self

    def load_state_dict(self, sd, strict=True):
        """Accepts the reference checkpoint naming: LLaMA weights (`model.layers.*`, `model.embed_tokens.weight`,
        `model.norm.weight`, `lm_head.weight`) and the non-LoRA trainables `model.mm_projector.*`,
        `model.region_extractor.*` (reference builder.py:64-79)."""
        proj = {k[len("model.mm_projector."):]: v for k, v in sd.items() if k.startswith("model.mm_projector.")}
        reg = {k[len("model.region_extractor."):]: v for k, v in sd.items() if k.startswith("model.region_extractor.")}
        llm = {k: v for k, v in sd.items() if not k.startswith(("model.mm_projector.", "model.region_extractor.",
                                                                "model.image_tower.", "model.video_tower."))}
        if proj and self.model.mm_projector is not None:
            self.model.mm_projector.load_state_dict(proj, strict)
        if reg and self.model.region_extractor is not None:
            self.model.region_extractor.load_state_dict(reg, strict)
        if llm:
            self._llama_sd = llm
            self.model.llama = None
        self.reset_prefix_cache()        # cached visual tokens / KV pages belong to the previous weights
        return [], []

    def resize_token_embeddings(self, new_num_tokens: Optional[int] = None):
        """PreTrainedModel.resize_token_embeddings as reference builder.py:146 calls it after tokenizer.add_tokens: embed_tokens
        and lm_head get `new_num_tokens` rows (old rows kept, new rows ZERO -- transformers draws them from N(0, 0.02^2); they are
        the untrained <im_patch> / <vi_patch> placeholders that never reach the model, a zero row gives them logit 0), and
        config.vocab_size follows."""
        if new_num_tokens is None:
            return self
        n = int(new_num_tokens)
        sd = self._llama_sd
        if sd is None:
            if self.model.llama is None:
                raise RuntimeError("resize_token_embeddings: load weights first")
            raise RuntimeError("resize_token_embeddings: call it before .to(device) packs the weights (load_pretrained_model does)")
        for key in ("model.embed_tokens.weight", "lm_head.weight"):
            w = sd[key]
            if w.shape[0] == n:
                continue
            out = torch.zeros((n, w.shape[1]), dtype=w.dtype, device=w.device)
            keep = min(n, w.shape[0])
            out[:keep] = w[:keep]
            sd[key] = out
        self.config.vocab_size = n
        self.vocab_size = n
        return self

    def to(self, device=None, dtype=None):
        """nn.Module.to: `dtype` (torch.bfloat16 / torch.float16) is the operand format the weights are packed in when they
        reach the GPU; it cannot change once the decoder is packed (the checkpoint copy is dropped then)."""
        if isinstance(device, torch.dtype):          # .to(torch.float16)
            device, dtype = None, device
        if dtype is not None:
            _lib.operand_of(dtype)
            if self.model.llama is not None and dtype != self.model.llama.dtype:
                raise RuntimeError(f"model weights are already packed as {self.model.llama.dtype}; load the checkpoint again to change the dtype")
            self._dtype = dtype
            for m in (self.model.mm_projector, self.model.region_extractor, self.model.image_tower, self.model.video_tower):
                if m is not None and hasattr(m, "to"):
                    m.to(dtype=dtype)
        if device is None:
            return self
        dev = torch.device(device)
        if dev.type == "cuda":
            if self._llama_sd is not None:
                self.model.llama = PackedLlama(self._llama_sd, self.config.to_dict(), dev, dtype=self._dtype)
                self._llama_sd = None
            for m in (self.model.mm_projector, self.model.region_extractor, self.model.image_tower, self.model.video_tower):
                if m is not None:
                    m.to(dev, dtype=self.dtype) if hasattr(m, "_dtype") else m.to(dev)
        self._device = dev
        return self

    def cuda(self):
        return self.to("cuda")

    def init_synthetic(self, device, seed=1234, vit_image: Optional[dict] = None, vit_video: Optional[dict] = None,
                       w_std=0.02, resize_for=None, dtype=None):
        """Random-init weights of the configured architecture, generated on `device` (no checkpoints exist offline). The
        values are drawn as bf16-representable numbers whatever `dtype` (the operand format they are packed in) is, so a
        bf16 and an fp16 model built from one seed hold the same weights (bf16 -> fp16 is exact above 2^-14)."""
        if dtype is not None:
            self._dtype = _resolve_dtype(dtype)
        gen = synth.make_generator(seed, device)
        cfg = self.config
        self._llama_sd = synth.llama_state(cfg.to_dict(), gen, device, w_std)
        if vit_image is not None:
            self.model.image_tower = build_image_tower(SimpleNamespace(mm_image_tower="synthetic/LanguageBind_Image",
                                                                       mm_vision_select_layer=cfg.mm_vision_select_layer), delay_load=True)
            self.model.image_tower._dtype = self._dtype
            self.model.image_tower.init_synthetic(vit_image, gen, device, w_std)
        if vit_video is not None:
            self.model.video_tower = build_video_tower(SimpleNamespace(mm_video_tower="synthetic/LanguageBind_Video_merge",
                                                                       mm_vision_select_layer=cfg.mm_vision_select_layer), delay_load=True)
            self.model.video_tower._dtype = self._dtype
            self.model.video_tower.init_synthetic(vit_video, gen, device, w_std)
        if self.model.mm_projector is None:
            self.model.mm_projector = build_vision_projector(cfg)
            self.model.region_extractor = build_region_extractor(cfg)
        for m in (self.model.mm_projector, self.model.region_extractor):
            if hasattr(m, "_dtype"):
                m._dtype = self._dtype
        self.model.mm_projector.init_synthetic(gen, device, w_std)
        self.model.region_extractor.init_synthetic(gen, device, w_std)
        if resize_for is not None:            # load_pretrained_model: tokenizer.add_tokens + resize_token_embeddings before packing
            n = resize_for()
            if n is not None:
                self.resize_token_embeddings(n)
        return self.to(device)

    # ---- precision modes -----------------------------------------------------------------------------------------
    def set_precise(self, level: int):
        """0: standard. 1: precise_qk -- prefills carry q / k (and the norm output behind their projection) as hi + lo operand pairs.
        2: a VERIFICATION mode -- every GEMM A operand of a prefill travels as an operand pair, from the towers' MLPs and the projector
        (whose output enters the decoder's fp32 residual stream unrounded) through all decoder GEMMs to the lm_head: about twice the
        GEMM work, and the fp16 build's full-depth logits end below north_star's 1e-3 of the reference's fp32 output (DESIGN.md 4,
        tests/test_gpu_parity_fulldepth.py). Needs the weights on the GPU (call after .to(device)); decode steps are unchanged."""
        level = int(level)
        llama = self.model.llama
        if llama is None:
            raise RuntimeError("set_precise: move the model to the GPU first (.to(device))")
        llama.set_precise(level)
        self.precise_level = level
        for tower in (self.get_image_tower(), self.get_video_tower()):
            if tower is not None and getattr(tower, "packed", None) is not None and hasattr(tower.packed, "set_precise"):
                tower.packed.set_precise(level)
        return self

    # ---- KV pool ----------------------------------------------------------------------------------------------
    def reset_prefix_cache(self):
        """Drop the kept conversation (KV pages go back to the pool) and the encoded-image cache."""
        if self._prefix is not None and self.kv is not None:
            self.kv.release(self._prefix.pages)
        self._prefix = None
        if self._vis_cache is not None:
            self._vis_cache.clear()

    def _ensure_kv(self, pages_needed: int):
        if self.kv is not None and len(self.kv.free) >= pages_needed:
            return
        if self._prefix is not None:            # the kept conversation is the first thing to go when pages run short
            if self.kv is not None:
                self.kv.release(self._prefix.pages)
            self._prefix = None
            if self.kv is not None and len(self.kv.free) >= pages_needed:
                return
        if self.kv is not None and len(self.kv.free) != self.kv.num_pages:
            raise RuntimeError("KV pool exhausted while sequences are live; release past_key_values or raise config.kv_pages")
        n = max(pages_needed, int(self.kv_pages or 0))
        self.kv = None
        self.kv = PagedKVCache(self.model.llama, n)

    # ---- reference llava_llama.py:57-102 ----------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None, inputs_embeds=None,
                labels=None, use_cache=None, output_attentions=None, output_hidden_states=None, images=None,
                regions=None, return_dict=None):
        splice = None
        if inputs_embeds is None:
            (input_ids, position_ids, attention_mask, past_key_values, inputs_embeds, labels) = \
                self.prepare_inputs_labels_for_multimodal(input_ids, position_ids, attention_mask, past_key_values,
                                                          labels, images, regions)
            splice = getattr(self, "_last_splice", None) if inputs_embeds is not None else None
        llama = self.model.llama
        if inputs_embeds is None:
            inputs_embeds = self.model.embed_tokens(input_ids)
        B, S, H = inputs_embeds.shape
        # valid rows of every sample (padding rows never enter the decoder: the batch is packed)
        if splice is not None:
            mask_host = splice[0]
        elif attention_mask is not None and past_key_values is None:
            mask_host = attention_mask.to(torch.int32).tolist()
        else:
            mask_host = [[1] * S for _ in range(B)]
        if past_key_values is None:
            lens = [sum(r) for r in mask_host]
            seqs = [SequenceState() for _ in range(B)]
            need = sum((l + 63) // 64 + 1 for l in lens)
            if use_cache:
                self._ensure_kv(need + B * 2)
            else:
                self._ensure_kv(need)
            past = PagedPast(self.kv, seqs, S)
            idx = [b * S + j for b in range(B) for j in range(S) if mask_host[b][j]]
        else:
            past = past_key_values
            seqs = past.seqs
            lens = [S] * B
            idx = list(range(B * S))
            past.padded_len += S
        flat = inputs_embeds.reshape(B * S, H)
        flat_lo = pair_lo(inputs_embeds)                 # precise level 2: the embeddings came out of the splice as an operand pair
        if flat_lo is not None:
            flat_lo = flat_lo.reshape(B * S, H)
        if len(idx) != B * S:   # pack the valid rows (padding never enters the decoder): a row gather by the splice kernel
            rowsel = torch.tensor(idx, dtype=torch.int32, device=flat.device)
            gather = torch.stack([torch.zeros_like(rowsel), rowsel], 1).contiguous()
            # (inputs_embeds of another float dtype are legal here: the gather kernel moves rows in the model's operand dtype)
            flat = ops.embed_splice(flat.to(llama.dtype).contiguous(), None, None, gather)
            if flat_lo is not None:
                flat_lo = ops.embed_splice(flat_lo.to(llama.dtype).contiguous(), None, None, gather)
        rows = flat.shape[0]
        if output_attentions:
            # (the reference's eager LlamaAttention returns the [B, heads, S, S] probabilities of every layer; the flash kernels never
            # materialise them -- refuse instead of returning None silently)
            raise NotImplementedError("output_attentions=True: the attention probabilities are never materialised by the HIP flash-attention kernels")
        trace = None
        if output_hidden_states:
            logits_p, hidden, trace = llama_forward(llama, past.kv, seqs, flat, lens, logit_rows=list(range(rows)), return_all_hidden=True,
                                                    embeds_lo=flat_lo)
        else:
            logits_p, hidden = llama_forward(llama, past.kv, seqs, flat, lens, logit_rows=list(range(rows)), return_hidden=True, embeds_lo=flat_lo)
        sel = torch.tensor(idx, device=flat.device) if len(idx) != B * S else None

        def unpack(t):       # packed rows [rows, D] -> the caller's padded [B, S, D] (zeros at the padding rows)
            if sel is None:
                return t.contiguous().view(B, S, t.shape[-1])
            full = torch.zeros((B * S, t.shape[-1]), dtype=t.dtype, device=t.device)
            full.index_copy_(0, sel, t.contiguous())
            return full.view(B, S, t.shape[-1])

        logits = unpack(logits_p)
        all_hidden = None
        if output_hidden_states:
            # transformers 4.31 LlamaModel.forward: the stream in front of every decoder layer (entry 0 = inputs_embeds), then the
            # FINAL-NORMED output of the last layer -- num_layers + 1 tensors [B, S, H]; fp32 here (the residual stream is fp32), the
            # last one in the operand dtype (it is the lm_head's operand)
            final = ops.rmsnorm(hidden, llama.final_norm, float(llama.model.rms_eps), dtype=llama.dtype)
            all_hidden = tuple(unpack(trace[l]) for l in range(trace.shape[0])) + (unpack(final),)
        loss = None
        if labels is not None:  # shifted cross entropy, ignore_index -100 (LlamaForCausalLM.forward)
            # one row per (sample, position < S-1): position j scores label j+1; the last position of every sample scores nothing
            shifted = torch.full((B, S), -100, dtype=torch.int32, device=logits.device)
            shifted[:, :-1] = labels[:, 1:].to(device=logits.device, dtype=torch.int32)
            loss = ops.cross_entropy(logits.view(B * S, llama.V), shifted.reshape(-1), ignore_index=-100)
        if not use_cache and past_key_values is None:
            past.release()
            past = None
        return CausalLMOutputWithPast(loss=loss, logits=logits, past_key_values=past, hidden_states=all_hidden, attentions=None)

    __call__ = forward

    # ---- reference llava_llama.py:104-114 ----------------------------------------------------------------------
    def prepare_inputs_for_generation(self, input_ids, past_key_values=None, inputs_embeds=None, **kwargs):
        images = kwargs.pop("images", None)
        regions = kwargs.pop("regions", None)
        if past_key_values is not None:
            input_ids = input_ids[:, -1:]
        inputs = {"input_ids": input_ids, "past_key_values": past_key_values, "use_cache": kwargs.get("use_cache", True),
                  "attention_mask": kwargs.get("attention_mask", None)}
        if inputs_embeds is not None and past_key_values is None:
            inputs = {"inputs_embeds": inputs_embeds, **{k: v for k, v in inputs.items() if k != "input_ids"}}
        if images is not None:
            inputs["images"] = images
        if regions is not None:
            inputs["regions"] = regions
        return inputs

    # ---- GenerationMixin.generate / sample, as app.py:562-571 drives it ----------------------------------------
    @torch.no_grad()
    def generate(self, input_ids=None, images=None, regions=None, attention_mask=None, do_sample=False,
                 temperature=1.0, top_p=1.0, top_k=None, max_new_tokens=None, max_length=None, use_cache=True,
                 stopping_criteria=None, eos_token_id=None, pad_token_id=None, num_beams=1, seed=None,
                 return_logits=False, padded_batch=False, **kwargs):
        """padded_batch=True (B > 1, right padding): reproduce what the REFERENCE's batched generate() returns -- transformers 4.31 reads the
        first token of every sample from the last COLUMN of the padded logits (a pad row for the shorter samples) and the decode-step
        fix-up (llava_arch.py:196-205) attends the pad rows of the spliced batch while masking the rows that share an index with the
        ids-length mask's zeros (engine.padded_batch_fixup; pinned by tests/golden/greedy_batch.npz). Default False: sequences are packed
        and every sample gets the ids the reference returns for it ALONE."""
        if num_beams != 1:
            raise NotImplementedError("beam search is not implemented (the reference entry points sample or go greedy)")
        dev = self.device
        input_ids = input_ids.to(dev)
        B = input_ids.shape[0]
        eos = self.config.eos_token_id if eos_token_id is None else eos_token_id
        eos_set = set(eos) if isinstance(eos, (list, tuple)) else ({eos} if eos is not None else set())
        pad = pad_token_id if pad_token_id is not None else (self.config.pad_token_id if self.config.pad_token_id is not None else 0)
        if max_new_tokens is None:
            max_new_tokens = (max_length - input_ids.shape[1]) if max_length else 20
        if top_k is None:     # GenerationConfig's default: the reference's sampling calls (app.py:562-571) run TopKLogitsWarper(50)
            top_k = int(getattr(self.config, "top_k", 50) or 0)
        # seed of the counter-based sampler: drawn from torch's global CPU generator (so torch.manual_seed makes sampling
        # reproducible, as with GenerationMixin.sample) and only when sampling -- greedy calls leave the RNG state alone
        sample_seed = 0
        if do_sample:
            sample_seed = int(seed) if seed is not None else int(torch.randint(0, 2 ** 31 - 1, (1,)).item())

        reuse = B == 1 and bool(getattr(self.config, "kv_prefix_reuse", True))
        cache = None
        if reuse and int(getattr(self.config, "vis_cache_entries", 16)) > 0:
            if self._vis_cache is None:
                from ...prefix_cache import VisualFeatureCache
                self._vis_cache = VisualFeatureCache(int(getattr(self.config, "vis_cache_entries", 16)))
            cache = self._vis_cache
        self._last_row_sig = None
        self.last_tower_items = None
        (_, _, _, _, embeds, _) = self.prepare_inputs_labels_for_multimodal(input_ids, None, attention_mask, None, None,
                                                                            images, regions, feature_cache=cache)
        llama = self.model.llama
        embeds_spliced = embeds is not None
        if embeds is None:
            embeds = self.model.embed_tokens(input_ids)
            mask_host = attention_mask.to(torch.int32).tolist() if attention_mask is not None else [[1] * input_ids.shape[1]] * B
        else:
            mask_host = self._last_splice[0]
        S, H = embeds.shape[1], embeds.shape[2]
        lens = [sum(r) for r in mask_host]
        idx = [b * S + j for b in range(B) for j in range(S) if mask_host[b][j]]
        flat = embeds.reshape(B * S, H)
        flat_lo = pair_lo(embeds)                        # precise level 2: the spliced embeddings are an operand pair
        if flat_lo is not None:
            flat_lo = flat_lo.reshape(B * S, H)
        if len(idx) != B * S:
            rowsel = torch.tensor(idx, dtype=torch.int32, device=dev)
            gather = torch.stack([torch.zeros_like(rowsel), rowsel], 1).contiguous()
            flat = ops.embed_splice(flat.to(llama.dtype).contiguous(), None, None, gather)
            if flat_lo is not None:
                flat_lo = ops.embed_splice(flat_lo.to(llama.dtype).contiguous(), None, None, gather)
        seqs = [SequenceState() for _ in range(B)]
        # ---- multi-turn KV reuse (batch 1): keep the pages of the longest common whole-page prefix of the last call ------------
        sig = None
        kept = 0
        if reuse:
            import numpy as np

            from ...prefix_cache import reusable_tokens
            if self._last_row_sig is not None:
                sig = self._last_row_sig[0][np.asarray(mask_host[0], dtype=bool)]
            elif embeds_spliced:
                # visual rows were spliced in without the feature cache (config.vis_cache_entries = 0): there is no per-row
                # signature of the spliced sequence, so this call neither reuses nor leaves a KV prefix
                reuse = False
                if self._prefix is not None:
                    self.reset_prefix_cache()
            else:
                sig = input_ids[0].cpu().numpy().astype(np.int64)[np.asarray(mask_host[0], dtype=bool)]
            if reuse and self._prefix is not None:
                kept = reusable_tokens(self._prefix.sig, sig) if self.kv is not None else 0
                seqs[0].pages = self._prefix.pages[:kept // 64]
                seqs[0].length = kept
                self.kv.release(self._prefix.pages[kept // 64:])
                self._prefix = None
        elif self._prefix is not None:
            self.reset_prefix_cache()
        pad_mode = bool(padded_batch) and B > 1
        if pad_mode and not embeds_spliced:
            # a text-only batch never reaches the reference's decode-step fix-up (llava_arch.py:196 returns early when images is
            # None); transformers then reads a shorter sample's first token from embed(pad_token_id) at position 1 -- another behaviour than
            # the one reproduced here (a ZERO row at position 0), so refuse instead of returning different ids silently
            raise NotImplementedError("generate(padded_batch=True) reproduces the reference's B > 1 result for MULTIMODAL (spliced) batches; a "
                                      "text-only batch takes another path in the reference (llava_arch.py:196-205) -- use the default packed batch")
        if pad_mode:
            am_rows = attention_mask.to(torch.int32).tolist() if attention_mask is not None else [[1] * input_ids.shape[1]] * B
            side = getattr(self.config, "tokenizer_padding_side", "right")
            if side == "left":
                # LEFT-padded batches (llava_arch.py:379-386): the first token comes from the last column, which is a real row; at a decode step
                # the fix-up (:196-205) extends the ids-length mask with ones, so sample i is denied the first p_ids[i] rows of its padded cache
                # (its padding in ID space) and gets position sum(mask) - 1. When every sample's padding in ID space equals its padding in
                # the SPLICED space -- the same number of <image> / <objs> rows in every sample, what a batched app.py turn looks like -- those
                # are exactly its pad rows and the position is its own length + step: the reference's ids ARE the ids each sample gets alone,
                # i.e. the packed batch below. Any other left-padded batch attends pad rows / loses real rows in the reference; not reproduced.
                if any(any(a > b_ for a, b_ in zip(r, r[1:])) for r in am_rows):
                    raise NotImplementedError("generate(padded_batch=True): with tokenizer_padding_side = 'left' the attention_mask must be left-padded (zeros, then ones)")
                p_ids = [len(r) - sum(r) for r in am_rows]
                p_sp = [S - l for l in lens]
                if p_ids != p_sp:
                    raise NotImplementedError(f"generate(padded_batch=True), left-padded: the samples' padding differs between the id space {p_ids} and the "
                                              f"spliced space {p_sp} (unequal numbers of visual rows): the reference then attends pad rows / masks real rows "
                                              "at decode steps (llava_arch.py:196-205); not reproduced -- use the default packed batch")
                pad_mode = False
            elif side != "right":
                raise NotImplementedError(f"generate(padded_batch=True): tokenizer_padding_side = {side!r}")
        if pad_mode:
            if any(any(a < b_ for a, b_ in zip(r, r[1:])) for r in am_rows):
                raise NotImplementedError("generate(padded_batch=True): attention_mask must be right-padded (ones, then zeros)")
            ids_valid = [sum(r) for r in am_rows]
        need = sum((l + max_new_tokens + 63) // 64 + 1 for l in lens) - kept // 64
        if pad_mode:     # pad rows join the cache, and the compaction needs a second set of pages for a moment
            need = sum(2 * ((S + 63) // 64 + 2) + (max_new_tokens + 63) // 64 + 1 for _ in lens)
        if self.kv is None or len(self.kv.free) < need:
            if kept:                                                          # cannot grow the pool around live pages: start over
                self.kv.release(seqs[0].pages)
                seqs[0].pages, seqs[0].length, kept = [], 0, 0
                need = sum((l + max_new_tokens + 63) // 64 + 1 for l in lens)
            self._ensure_kv(need)
        self.last_generate_stats = {"prompt_rows": int(sum(lens)), "reused_tokens": int(kept),
                                    "prefill_rows": int(sum(lens) - kept), "tower_items": self.last_tower_items}
        if kept:
            flat, lens = flat[kept:], [lens[0] - kept]
            flat_lo = None if flat_lo is None else flat_lo[kept:]
        # ---- decode loop: device-resident step state, the next pass is enqueued before the host has seen the token it consumes
        L0 = input_ids.shape[1]
        out_host = torch.empty((B, L0 + max_new_tokens), dtype=torch.long)
        out_host[:, :L0] = input_ids.cpu()
        n_out = L0
        all_logits = []
        state = None
        try:   # (the prefill is inside: pages it took before a failure go back to the pool / the kept prefix below)
            logits = llama_forward(llama, self.kv, seqs, flat, lens, embeds_lo=flat_lo)     # [B, V]: last position of every sequence
            if pad_mode:
                from ...engine import padded_batch_fixup
                fix = padded_batch_fixup(llama, self.kv, seqs, S, ids_valid, input_ids.shape[1])
                logits = logits.clone()
                for b, lg in fix["pad_logits"].items():       # shorter samples: the first token comes from their last PAD row
                    logits[b] = lg
                self.last_generate_stats["padded_batch"] = {"pad_rows": {b: S - lens[b] for b in fix["pad_logits"]}, "masked_rows": fix["holes"]}
            for step in range(max_new_tokens):
                if return_logits:
                    all_logits.append(logits.clone())
                if do_sample:   # device-side temperature / top-k / top-p sampler (no host sync, counter-based RNG)
                    nxt = ops.sample_top_p(logits, temperature, top_p if top_p else 1.0, sample_seed, step, top_k=int(top_k or 0))
                else:
                    nxt = ops.argmax(logits)
                last = step + 1 == max_new_tokens
                scores = logits
                if state is None and last:    # single-token call (prefill benchmarks, scoring): no decode state to set up
                    toks = nxt.tolist()
                    fin = [t in eos_set for t in toks]
                else:
                    if state is None:
                        state = DecodeState(llama, self.kv, seqs, max_new_tokens, sorted(eos_set), pad)
                    slot = state.feed(nxt)
                    if not last:
                        logits = state.forward()          # speculative: rolled back below if this token ends the run
                    toks, fin = state.read(slot)
                out_host[:, n_out] = torch.tensor(toks, dtype=torch.long)
                n_out += 1
                stop = all(fin)
                if not stop and stopping_criteria is not None:
                    stop = any(c(out_host[:, :n_out], scores) for c in stopping_criteria)    # StoppingCriteriaList: any()
                if stop:
                    if state is not None and not last:
                        state.rollback()
                    break
        finally:
            if reuse and sig is not None and seqs[0].length > 0:
                import numpy as np

                from ...prefix_cache import PrefixKV
                gen_ids = out_host[0, L0:n_out].numpy().astype(np.int64)
                full = np.concatenate([sig, gen_ids])[:seqs[0].length]      # the last sampled token was never fed back
                self._prefix = PrefixKV(full, seqs[0].pages)
            else:
                for s in seqs:
                    self.kv.release(s.pages)
        out = out_host[:, :n_out].to(dev)
        if return_logits:
            return out, all_logits
        return out


VitronLlamaForCausalLM = LlavaLlamaForCausalLM  # name used by BASELINE.json's north_star; the reference class is LlavaLlamaForCausalLM
