"""diffusers.models.attention_processor (0.30.2): Attention + AttnProcessor2_0 (the default when torch has SDPA)."""
import torch
import torch.nn.functional as F
from torch import nn

from ._placeholder import placeholder


class AttnProcessor2_0:
    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None, *args, **kwargs):
        residual = hidden_states
        if attn.spatial_norm is not None:
            hidden_states = attn.spatial_norm(hidden_states, temb)
        input_ndim = hidden_states.ndim
        if input_ndim == 4:
            batch_size, channel, height, width = hidden_states.shape
            hidden_states = hidden_states.view(batch_size, channel, height * width).transpose(1, 2)
        batch_size, sequence_length, _ = hidden_states.shape if encoder_hidden_states is None else encoder_hidden_states.shape
        if attention_mask is not None:
            attention_mask = attn.prepare_attention_mask(attention_mask, sequence_length, batch_size)
            attention_mask = attention_mask.view(batch_size, attn.heads, -1, attention_mask.shape[-1])
        if attn.group_norm is not None:
            hidden_states = attn.group_norm(hidden_states.transpose(1, 2)).transpose(1, 2)
        query = attn.to_q(hidden_states)
        if encoder_hidden_states is None:
            encoder_hidden_states = hidden_states
        elif attn.norm_cross:
            encoder_hidden_states = attn.norm_encoder_hidden_states(encoder_hidden_states)
        key = attn.to_k(encoder_hidden_states)
        value = attn.to_v(encoder_hidden_states)
        inner_dim = key.shape[-1]
        head_dim = inner_dim // attn.heads
        query = query.view(batch_size, -1, attn.heads, head_dim).transpose(1, 2)
        key = key.view(batch_size, -1, attn.heads, head_dim).transpose(1, 2)
        value = value.view(batch_size, -1, attn.heads, head_dim).transpose(1, 2)
        hidden_states = F.scaled_dot_product_attention(query, key, value, attn_mask=attention_mask, dropout_p=0.0, is_causal=False)
        hidden_states = hidden_states.transpose(1, 2).reshape(batch_size, -1, attn.heads * head_dim)
        hidden_states = hidden_states.to(query.dtype)
        hidden_states = attn.to_out[0](hidden_states)
        hidden_states = attn.to_out[1](hidden_states)
        if input_ndim == 4:
            hidden_states = hidden_states.transpose(-1, -2).reshape(batch_size, channel, height, width)
        if attn.residual_connection:
            hidden_states = hidden_states + residual
        hidden_states = hidden_states / attn.rescale_output_factor
        return hidden_states


AttnProcessor = AttnProcessor2_0
AttnAddedKVProcessor = placeholder("AttnAddedKVProcessor")
AttnAddedKVProcessor2_0 = placeholder("AttnAddedKVProcessor2_0")
AttentionProcessor = object
ADDED_KV_ATTENTION_PROCESSORS = (AttnAddedKVProcessor, AttnAddedKVProcessor2_0)
CROSS_ATTENTION_PROCESSORS = (AttnProcessor2_0,)


class Attention(nn.Module):
    def __init__(self, query_dim, cross_attention_dim=None, heads=8, kv_heads=None, dim_head=64, dropout=0.0, bias=False,
                 upcast_attention=False, upcast_softmax=False, cross_attention_norm=None, cross_attention_norm_num_groups=32, qk_norm=None,
                 added_kv_proj_dim=None, added_proj_bias=True, norm_num_groups=None, spatial_norm_dim=None, out_bias=True, scale_qk=True,
                 only_cross_attention=False, eps=1e-5, rescale_output_factor=1.0, residual_connection=False,
                 _from_deprecated_attn_block=False, processor=None, out_dim=None, context_pre_only=None, pre_only=False):
        super().__init__()
        assert cross_attention_norm is None and qk_norm is None and added_kv_proj_dim is None and spatial_norm_dim is None and kv_heads is None
        self.inner_dim = out_dim if out_dim is not None else dim_head * heads
        self.query_dim = query_dim
        self.is_cross_attention = cross_attention_dim is not None
        self.cross_attention_dim = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.upcast_attention, self.upcast_softmax = upcast_attention, upcast_softmax
        self.rescale_output_factor, self.residual_connection = rescale_output_factor, residual_connection
        self._from_deprecated_attn_block = _from_deprecated_attn_block
        self.scale = dim_head ** -0.5 if scale_qk else 1.0
        self.heads = out_dim // dim_head if out_dim is not None else heads
        self.only_cross_attention = only_cross_attention
        self.group_norm = nn.GroupNorm(num_channels=query_dim, num_groups=norm_num_groups, eps=eps, affine=True) if norm_num_groups is not None else None
        self.spatial_norm = None
        self.norm_cross = None
        self.to_q = nn.Linear(query_dim, self.inner_dim, bias=bias)
        self.to_k = nn.Linear(self.cross_attention_dim, self.inner_dim, bias=bias)
        self.to_v = nn.Linear(self.cross_attention_dim, self.inner_dim, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(self.inner_dim, query_dim, bias=out_bias), nn.Dropout(dropout)])
        self.processor = processor if processor is not None else AttnProcessor2_0()

    def set_use_memory_efficient_attention_xformers(self, use_memory_efficient_attention_xformers, attention_op=None):
        # diffusers would install XFormersAttnProcessor (same mathematics as SDPA); the reference's CustomJointAttention overrides this
        self.processor = AttnProcessor2_0()

    def set_processor(self, processor, _remove_lora=False):
        self.processor = processor

    def get_processor(self, return_deprecated_lora=False):
        return self.processor

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **cross_attention_kwargs):
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states, attention_mask=attention_mask,
                              **cross_attention_kwargs)

    def batch_to_head_dim(self, tensor):
        head_size = self.heads
        batch_size, seq_len, dim = tensor.shape
        tensor = tensor.reshape(batch_size // head_size, head_size, seq_len, dim)
        return tensor.permute(0, 2, 1, 3).reshape(batch_size // head_size, seq_len, dim * head_size)

    def head_to_batch_dim(self, tensor, out_dim=3):
        head_size = self.heads
        batch_size, seq_len, dim = tensor.shape
        tensor = tensor.reshape(batch_size, seq_len, head_size, dim // head_size).permute(0, 2, 1, 3)
        if out_dim == 3:
            tensor = tensor.reshape(batch_size * head_size, seq_len, dim // head_size)
        return tensor

    def prepare_attention_mask(self, attention_mask, target_length, batch_size, out_dim=3):
        if attention_mask is None:
            return attention_mask
        raise NotImplementedError("attention masks are not on the E2E-FT path")
