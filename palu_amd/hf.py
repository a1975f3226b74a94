"""Whole-model decode with latent caches: the bridge between `LlamaPaluAttention` (which speaks the transformers-4.37.2
cache protocol the reference pins: `get_usable_length` / `update`, kernel/palu_attention.py:147-263) and the transformers
release installed here (5.x: `past_key_values=Cache`, `position_embeddings`, mask built by `create_causal_mask`).

SURVEY.md 8(f) N2: in the reference the L4 models (`palu/model/svd_llama/modeling_palu_llama.py:7-35`) reconstruct full
K/V and never reach the kernel path; this module is what lets a whole `LlamaForCausalLM` decode through the HIP step:

    model = convert_llama_to_palu(model, rank_k=1024, rank_v=3072, group_size=4)      # per layer: from_attention + adapter
    cache = PaluCacheHF(bits=16)                                                      # or bits=4 / 3: packed latents
    out = model(input_ids, past_key_values=cache, use_cache=True)                     # prompt pass fills the latent caches
    out = model(next_token, past_key_values=cache, use_cache=True)                    # one HIP decode step per layer

`PaluCacheHF` is a `transformers.Cache` (so `get_seq_length` / `get_mask_sizes` serve HF's mask and position logic) that
carries one `LatentCache` or `QuantLatentCache` for all layers; K/V never exist in reconstructed form.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn

from .kernel.palu_attention import LatentCache, LlamaPaluAttention, QuantLatentCache, additive_mask

try:                                   # transformers is an optional dependency of this bridge only
    from transformers.cache_utils import Cache as _HFCache
except Exception:                      # noqa: BLE001
    _HFCache = object


class PaluCacheHF(_HFCache):
    """`transformers.Cache` facade over the latent caches of every layer.  The attention modules never call `update`
    with reconstructed K/V (there are none): they append latent rows through `self.latent`."""

    def __init__(self, bits: int = 16, capacity: int = 0, headroom: int = 256, group_size: int = 0):
        """group_size: quantize_tensor's column-group width for packed caches (quant.py:11-13, `--lt_group_size`); 0 = one
        (scale, zero) pair per (token, head-group) row, the reference default."""
        if _HFCache is not object:
            try:
                super().__init__(layers=[])
            except TypeError:          # older Cache.__init__ without arguments
                super().__init__()
        self.latent = (LatentCache(capacity, headroom) if bits >= 16
                       else QuantLatentCache(bits, capacity, headroom, group_size=group_size))
        self.bits, self._capacity, self._headroom, self._group_size = bits, capacity, headroom, group_size
        self._mask_memo = None            # (mask identity, "is the plain causal mask") of the current forward pass
        self._std_positions = None        # True / False: recorded by a prompt pass (or assume_standard_positions); None: unknown

    # -- what transformers' model code asks a cache ------------------------------------------------------------
    def get_seq_length(self, layer_idx: int = 0) -> int:
        return self.latent.get_seq_length(layer_idx)

    def get_mask_sizes(self, query_length, layer_idx: int = 0):
        q = int(query_length.shape[0]) if isinstance(query_length, torch.Tensor) else int(query_length)
        return self.get_seq_length(layer_idx) + q, 0

    def get_max_cache_shape(self, layer_idx: int = 0) -> int:
        return -1

    def get_max_length(self):
        return None

    @property
    def is_compileable(self) -> bool:
        return False

    def update(self, key_states, value_states, layer_idx, cache_kwargs=None):
        raise RuntimeError("PaluCacheHF holds LATENT rows: it is updated by LlamaPaluAttention, not with reconstructed K/V")

    def assume_standard_positions(self, flag: bool = True):
        """For a cache filled WITHOUT a prompt pass through the model (load_cache, synthetic rows): declare that the decode
        positions the caller will pass are the rows they append at (position == cache length).  Decode steps then never read
        `position_ids` back from the device (a host sync per layer and token; illegal under graph capture).  A prompt pass
        records this by itself; without either, device `position_ids` are passed through to the module."""
        self._std_positions = bool(flag)

    def reset(self):
        self.latent = (LatentCache(self._capacity, self._headroom) if self.bits >= 16
                       else QuantLatentCache(self.bits, self._capacity, self._headroom, group_size=self._group_size))
        self._mask_memo = None
        self._std_positions = None          # recorded by the next prompt pass (PaluAttentionHF.forward)

    def __len__(self):
        return len(getattr(self.latent, "_state", {})) if hasattr(self.latent, "_state") else 0


class PaluAttentionHF(nn.Module):
    """`LlamaPaluAttention` behind the forward signature transformers 5.x decoder layers call
    (`hidden_states, position_embeddings, attention_mask, past_key_values, position_ids, ...`) -> (output, weights)."""

    def __init__(self, inner: LlamaPaluAttention):
        super().__init__()
        self.inner = inner
        self.layer_idx = inner.layer_idx
        self.config = inner.config

    def forward(self, hidden_states: torch.Tensor, position_embeddings=None, attention_mask: Optional[torch.Tensor] = None,
                past_key_values=None, position_ids: Optional[torch.LongTensor] = None, **kwargs):
        cache = past_key_values.latent if isinstance(past_key_values, PaluCacheHF) else past_key_values
        q_len = hidden_states.shape[1]
        is_causal = None
        orig = attention_mask
        # One conversion + one "is this the plain causal mask" decision per FORWARD PASS, not per layer: the memo lives on
        # the cache object and is keyed on the IDENTITY of the mask tensor transformers hands to every layer, which it holds
        # a reference to (a key made of data_ptr / shape / version could be inherited by a different mask that
        # the caching allocator placed at the freed address).
        memo = getattr(past_key_values, "_mask_memo", None)
        if memo is not None and (orig is None or memo[0] is not orig or memo[3] != orig._version):
            # an entry of another pass (the prompt's masks: [1, 1, q, kv] bool + fp16, hundreds of MiB at 8k tokens) is not kept
            memo = None
            try:
                past_key_values._mask_memo = None
            except AttributeError:
                pass
        last_layer = self.layer_idx == getattr(self.config, "num_hidden_layers", -1) - 1
        if orig is not None and memo is not None:
            attention_mask, is_causal = memo[1], memo[2]
            if last_layer:
                past_key_values._mask_memo = None
        else:
            if attention_mask is not None and attention_mask.dtype == torch.bool:
                # transformers 5.x (sdpa / create_causal_mask) hands over BOOLEAN masks, True = attend: the module speaks the
                # reference's additive convention (kernel/palu_attention.py:229-234), so convert first -- a bool mask cast
                # to fp16 would add +1 to attended positions and mask nothing
                attention_mask = additive_mask(attention_mask, hidden_states.dtype)
            if q_len > 1 and attention_mask is not None:
                # a mask tensor may carry padding: decide whether it is the plain causal one (flash kernel) or not (general
                # path).  The test syncs; a mask whose width is not past + q_len is left to the module's own shape check.
                past = cache.get_seq_length(self.layer_idx) if cache is not None else 0
                is_causal = (attention_mask.shape[0] == 1 and attention_mask.shape[-1] == past + q_len
                             and attention_mask.shape[-2] == q_len
                             and bool(self.inner._mask_is_causal(attention_mask, q_len, past)))
            if orig is not None and past_key_values is not None:
                try:
                    # (identity AND version: a mask tensor mutated in place and reused must be converted again; the last layer
                    # of the pass drops the entry.  Decode steps memoise too: converting a [1, 1, 1, kv] mask is three small
                    # launches per LAYER otherwise -- 7 us of a 220 us layer, measured in tools/bench_model.py)
                    past_key_values._mask_memo = None if last_layer else (orig, attention_mask, is_causal, orig._version)
                except AttributeError:
                    pass
        if q_len == 1:
            # one token attends to the whole cache.  The mask (if any) is passed through as it is: testing it for "all
            # zeros" would be a device-to-host sync per layer and token, and is illegal under graph capture; an all-zero
            # mask only selects the masked softmax kernel, it changes no result.
            # The token's position is the row its latents are appended at (cache rows ARE absolute positions,
            # kernel/abx_rope.py:114-150: key position = row index), so the module derives it from the cache length: reading
            # it out of the device tensor transformers hands over would be a host sync per layer and token (and raises
            # inside a graph capture).
            # That is only right when the caller's position IS the cache length (the standard case).  Left-padded or offset
            # positions must be passed through (the reference rotates the decode query with position_ids,
            # kernel/palu_attention.py:218-219): a CPU tensor is compared on the host; for a device tensor the prompt pass
            # recorded whether its positions started at the cache length (`_std_positions`), and only then it is dropped.
            if position_ids is not None:
                clen = cache.get_seq_length(self.layer_idx) if cache is not None else 0
                if position_ids.device.type == "cpu":
                    if int(position_ids.reshape(-1)[0]) == clen:
                        position_ids = None
                elif getattr(past_key_values, "_std_positions", None) is True:
                    position_ids = None            # (None = nothing recorded, e.g. a cache filled by load_cache: pass them through)
            is_causal = None
        elif attention_mask is None:
            is_causal = True               # the model is causal; without a mask tensor the module would apply none (:229)
        if q_len > 1 and past_key_values is not None and self.layer_idx == 0:
            # (one host read per PROMPT pass, first layer only: are this pass's positions the rows it appends at, i.e.
            #  arange(cache length, cache length + q_len)?  Judged from the WHOLE tensor -- left padding keeps element 0 at the
            #  cache length often enough -- and recorded anew by every prompt pass, also one without position_ids: standard)
            try:
                past0 = cache.get_seq_length(0) if cache is not None else 0
                if position_ids is None:
                    past_key_values._std_positions = True
                else:
                    pid = position_ids.reshape(-1)
                    want = torch.arange(past0, past0 + pid.numel(), device=pid.device, dtype=pid.dtype)
                    past_key_values._std_positions = bool(pid.numel() == q_len and torch.equal(pid, want))
            except AttributeError:
                pass
        out, weights, _ = self.inner(hidden_states, attention_mask=attention_mask, position_ids=position_ids,
                                     past_key_value=cache, output_attentions=bool(kwargs.get("output_attentions", False)),
                                     is_causal=is_causal)
        return out, weights


def palu_config_from(config, rank_k: int, rank_v: int, group_size: int):
    """The config fields LlamaPaluAttention reads (kernel/palu_attention.py:132-145) on top of the model's own config."""
    import copy
    c = copy.copy(config)
    heads = config.num_attention_heads
    kv = getattr(config, "num_key_value_heads", None) or heads
    # GQA checkpoints: `group_size` counts KEY/VALUE heads per low-rank group, as in the reference's own GQA wrapper
    # (palu/model/svd_mistral/modeling_palu_mistral.py:37-59, get_kv_info: num_lr_groups = num_key_value_heads // group)
    if heads % kv or kv % group_size:
        raise ValueError(f"num_key_value_heads ({kv}) must divide num_attention_heads ({heads}) and be divisible by "
                         f"group_size ({group_size})")
    c.group_size = group_size
    c.num_groups = kv // group_size
    c.total_rank_k, c.total_rank_v = rank_k, rank_v
    if not hasattr(c, "attention_bias"):
        c.attention_bias = False
    return c


@torch.no_grad()
def convert_llama_to_palu(model, rank_k: int, rank_v: int, group_size: int = 4, hadamard: bool = False):
    """Replace every `self_attn` of a `LlamaForCausalLM` / `LlamaModel` by the low-rank module
    (`LlamaPaluAttention.from_attention`: group-wise SVD of k/v projections, U_v folded into o_proj) behind the
    transformers-5 adapter.  Returns the same model object."""
    base = getattr(model, "model", model)
    cfg = palu_config_from(base.config, rank_k, rank_v, group_size)
    for layer in base.layers:
        att = layer.self_attn
        if isinstance(att, PaluAttentionHF):
            continue
        inner = LlamaPaluAttention.from_attention(att, cfg)
        inner.layer_idx = getattr(att, "layer_idx", inner.layer_idx)
        inner = inner.to(device=att.q_proj.weight.device, dtype=att.q_proj.weight.dtype)
        if hadamard:
            inner.fuse_hadamard()
        inner.prepare_decode()            # fragments + shared-B decision now, not inside the first (possibly captured) step
        layer.self_attn = PaluAttentionHF(inner)
    return model


class _DecodeGemvLinear(nn.Module):
    """An `nn.Linear` whose batch-1, one-token call is the HIP GEMV (weights streamed once at HBM rate, `palu_gemv_bias_f16`);
    every other call is the wrapped linear.  Whole-model decode only -- the attention module has its own."""

    def __init__(self, lin: nn.Linear):
        super().__init__()
        self.lin = lin
        w = lin.weight
        self._ok = (w.dtype == torch.float16 and w.is_cuda and w.stride(1) == 1 and w.shape[1] % 8 == 0
                    and w.shape[1] * 2 <= 64 * 1024 and w.stride(0) % 8 == 0)

    @property
    def weight(self):
        return self.lin.weight

    def forward(self, x):
        w = self.lin.weight
        if not (self._ok and x.numel() == w.shape[1] and x.dtype == torch.float16 and x.is_cuda and x.is_contiguous()):
            return self.lin(x)
        from . import _lib
        y = torch.empty(x.shape[:-1] + (w.shape[0],), dtype=x.dtype, device=x.device)
        b = self.lin.bias
        _lib.check(_lib.lib.palu_gemv_bias_f16(w.data_ptr(), w.stride(0), x.data_ptr(), 0 if b is None else b.data_ptr(),
                                               y.data_ptr(), w.shape[0], w.shape[1], _lib.current_stream()), "palu_gemv_bias_f16")
        return y


class _DecodeGatedMLP(nn.Module):
    """transformers' LlamaMLP (`down_proj(act_fn(gate_proj(x)) * up_proj(x))`) whose one-token call is two HIP launches: the
    gate and up GEMVs with the SiLU product in their epilogue (`palu_gemv_silu_mul_f16`), then the down GEMV."""

    def __init__(self, mlp):
        super().__init__()
        self.mlp = mlp
        g, u, d = mlp.gate_proj, mlp.up_proj, mlp.down_proj
        act = getattr(mlp, "act_fn", None)
        silu = isinstance(act, nn.SiLU) or getattr(act, "__name__", "") == "silu" or type(act).__name__ in ("SiLUActivation", "SiLU")
        self._ok = (silu and all(l.bias is None and l.weight.dtype == torch.float16 and l.weight.is_cuda and l.weight.stride(1) == 1
                                 and l.weight.shape[1] % 8 == 0 and l.weight.shape[1] * 2 <= 64 * 1024 for l in (g, u, d))
                    and g.weight.shape == u.weight.shape)

    def forward(self, x):
        g, u, d = self.mlp.gate_proj.weight, self.mlp.up_proj.weight, self.mlp.down_proj.weight
        if not (self._ok and x.numel() == g.shape[1] and x.dtype == torch.float16 and x.is_cuda and x.is_contiguous()):
            return self.mlp(x)
        from . import _lib
        s = _lib.current_stream()
        act = torch.empty(g.shape[0], dtype=x.dtype, device=x.device)
        _lib.check(_lib.lib.palu_gemv_silu_mul_f16(g.data_ptr(), g.stride(0), u.data_ptr(), u.stride(0), x.data_ptr(),
                                                   act.data_ptr(), g.shape[0], g.shape[1], s), "palu_gemv_silu_mul_f16")
        y = torch.empty(x.shape[:-1] + (d.shape[0],), dtype=x.dtype, device=x.device)
        _lib.check(_lib.lib.palu_gemv_f16(d.data_ptr(), d.stride(0), act.data_ptr(), y.data_ptr(), d.shape[0], d.shape[1], s),
                   "palu_gemv_f16")
        return y


class _DecodeRMSNorm(nn.Module):
    """LlamaRMSNorm whose one-token call is one HIP launch (`palu_rmsnorm_row_f16`) instead of seven torch kernels."""

    def __init__(self, norm):
        super().__init__()
        self.norm = norm
        w = norm.weight
        self.eps = float(getattr(norm, "variance_epsilon", getattr(norm, "eps", 1e-6)))
        self._ok = w.dtype == torch.float16 and w.is_cuda and w.dim() == 1 and w.shape[0] % 8 == 0 and w.is_contiguous()

    @property
    def weight(self):
        return self.norm.weight

    def forward(self, x):
        w = self.norm.weight
        if not (self._ok and x.numel() == w.shape[0] and x.dtype == torch.float16 and x.is_cuda and x.is_contiguous()):
            return self.norm(x)
        from . import _lib
        y = torch.empty_like(x)
        _lib.check(_lib.lib.palu_rmsnorm_row_f16(x.data_ptr(), w.data_ptr(), y.data_ptr(), w.shape[0], self.eps,
                                                 _lib.current_stream()), "palu_rmsnorm_row_f16")
        return y


def use_hip_decode_linears(model):
    """Route the one-token calls of every gated MLP, every RMSNorm and `lm_head` of a Llama-style model through HIP kernels
    (prompt passes and batches keep the wrapped torch modules).  Optional: whole-model decode throughput, not attention
    parity."""
    base = getattr(model, "model", model)
    for layer in base.layers:
        if hasattr(layer, "mlp") and not isinstance(layer.mlp, _DecodeGatedMLP) and all(
                hasattr(layer.mlp, a) for a in ("gate_proj", "up_proj", "down_proj")):
            layer.mlp = _DecodeGatedMLP(layer.mlp)
        for name in ("input_layernorm", "post_attention_layernorm"):
            n = getattr(layer, name, None)
            if n is not None and not isinstance(n, _DecodeRMSNorm) and type(n).__name__.endswith("RMSNorm"):
                setattr(layer, name, _DecodeRMSNorm(n))
    fin = getattr(base, "norm", None)
    if fin is not None and not isinstance(fin, _DecodeRMSNorm) and type(fin).__name__.endswith("RMSNorm"):
        base.norm = _DecodeRMSNorm(fin)
    head = getattr(model, "lm_head", None)
    if isinstance(head, nn.Linear):
        model.lm_head = _DecodeGemvLinear(head)
    return model
