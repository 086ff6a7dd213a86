import torch.nn.functional as F


def memory_efficient_attention(query, key, value, attn_bias=None, p=0.0, scale=None):
    """[B*heads, N, d] tensors, softmax(q k^T / sqrt(d)) v  (GeoWizard/geowizard/models/attention.py:497 call site)"""
    assert p == 0.0
    return F.scaled_dot_product_attention(query, key, value, attn_mask=attn_bias, scale=scale)
