def maybe_allow_in_graph(cls):
    return cls


def apply_freeu(resolution_idx, hidden_states, res_hidden_states, **freeu_kwargs):
    raise NotImplementedError("FreeU is not on the E2E-FT path")
