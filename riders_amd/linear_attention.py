"""LoFTR-style linear-attention transformer on MI355X.  Same classes / signatures / state_dict keys as the
reference's RCNet/linear_attention.py (elu_feature_map :7, LinearAttention :12, LoFTREncoderLayer :84,
LocalFeatureTransformer :139); FullAttention is never instantiated on this path and is not provided.

Projections and the MLP are MFMA GEMMs (1x1 instances of the implicit-GEMM conv kernel, the [x, message]
concat folded into the gather); phi(Q)(phi(K)^T V) is rd_linear_attention_{fwd,bwd}; LayerNorm + residual
are fused row kernels.
"""
import copy

import torch
import torch.nn as nn

from . import engine
from .engine import ACT_NONE, ACT_RELU


class LinearAttention(nn.Module):
    """Reference: RCNet/linear_attention.py:12-45.  queries [N,L,H,D], keys/values [N,S,H,D] -> [N,L,H,D]."""

    def __init__(self, eps=1e-6):
        super().__init__()
        self.eps = eps

    def _fwd(self, q, k, v, N, L, S, H):
        return engine.linear_attention(q, k, v, N, L, S, H, self.eps)

    def forward(self, queries, keys, values, q_mask=None, kv_mask=None):
        if q_mask is not None or kv_mask is not None:
            raise NotImplementedError('masks are always None on the RIDERS path (networks.py:444)')
        N, L, H, D = queries.shape
        S = keys.shape[1]

        def run(q, k, v):
            o = self._fwd(engine.tokens_in(q, H * D), engine.tokens_in(k, H * D), engine.tokens_in(v, H * D), N, L, S, H)
            return engine.alias_cast_view(o, q.dtype, (N, L, H, D))
        return engine.run_region(run, (queries, keys, values), [])


class LoFTREncoderLayer(nn.Module):
    """Reference: RCNet/linear_attention.py:84-135."""

    def __init__(self, d_model, nhead, attention='linear'):
        super(LoFTREncoderLayer, self).__init__()
        if attention != 'linear':
            raise NotImplementedError("only attention='linear' is used (linear_attention.py:142, networks.py:378)")
        self.dim = d_model // nhead
        self.nhead = nhead
        self.q_proj = nn.Linear(d_model, d_model, bias=False)
        self.k_proj = nn.Linear(d_model, d_model, bias=False)
        self.v_proj = nn.Linear(d_model, d_model, bias=False)
        self.attention = LinearAttention()
        self.merge = nn.Linear(d_model, d_model, bias=False)
        self.mlp = nn.Sequential(
            nn.Linear(d_model * 2, d_model * 2, bias=False),
            nn.ReLU(True),
            nn.Linear(d_model * 2, d_model, bias=False),
        )
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)

    def _fwd(self, x, source, N, L, S):
        """x (N*L, C), source (N*S, C) token matrices -> (N*L, C)."""
        if engine.fused_loftr() and x.shape[1] == 128 and self.nhead == 8 and L <= 32 and S <= 32:
            return engine.loftr_layer(x, source, self, N, L, S)      # one workgroup per sequence walks the whole layer
        q = engine.linear(x, self.q_proj.weight)
        k = engine.linear(source, self.k_proj.weight)
        v = engine.linear(source, self.v_proj.weight)
        message = self.attention._fwd(q, k, v, N, L, S, self.nhead)
        message = engine.linear(message, self.merge.weight)
        message = engine.layernorm(message, self.norm1)
        hidden = engine.linear(x, self.mlp[0].weight, x2=message, act=ACT_RELU)   # mlp(cat([x, message]))
        message = engine.linear(hidden, self.mlp[2].weight)
        return engine.layernorm(message, self.norm2, residual=x)                     # x + norm2(message)

    def forward(self, x, source, x_mask=None, source_mask=None):
        if x_mask is not None or source_mask is not None:
            raise NotImplementedError('masks are always None on the RIDERS path')
        N, L, C = x.shape
        S = source.shape[1]
        same = source is x

        def run(x, source=None):
            xt = engine.tokens_in(x, C)
            st = xt if source is None else engine.tokens_in(source, C)
            out = self._fwd(xt, st, N, L, S)
            return engine.alias_cast_view(out, x.dtype, (N, L, C))
        ins = (x,) if same else (x, source)
        return engine.run_region(run, ins, list(self.parameters()))


class LocalFeatureTransformer(nn.Module):
    """Reference: RCNet/linear_attention.py:139-184.  layer_names = type * n_layers; 'cross' feeds the already
    updated feat0 into feat1's update with the same layer weights."""

    def __init__(self, type, n_layers=1, d_model=256, nhead=8, attention='linear'):
        super(LocalFeatureTransformer, self).__init__()
        self.d_model = d_model
        self.nhead = nhead
        self.layer_names = type * n_layers
        self.attention = attention
        encoder_layer = LoFTREncoderLayer(self.d_model, self.nhead, self.attention)
        self.layers = nn.ModuleList([copy.deepcopy(encoder_layer) for _ in range(len(self.layer_names))])
        self._reset_parameters()

    def _reset_parameters(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def _fwd(self, f0, f1, N, L, S):
        fused = engine.fused_loftr() and f0.shape[1] == 128 and L == S and L <= 32 and all(l.nhead == 8 for l in self.layers)
        if fused and engine._state.get("merge_self_layers", True):
            return self._fwd_merged(f0, f1, N, L)
        for layer, name in zip(self.layers, self.layer_names):
            if name == 'self':
                f0 = layer._fwd(f0, f0, N, L, L)
                f1 = layer._fwd(f1, f1, N, S, S)
            elif name == 'cross':
                f0 = layer._fwd(f0, f1, N, L, S)
                f1 = layer._fwd(f1, f0, N, S, L)
            else:
                raise KeyError
        return f0, f1

    def _fwd_merged(self, f0, f1, N, L):
        """Both token streams live in one (2 N L, C) matrix.  A 'self' layer applies the same weights to each stream independently
        (linear_attention.py:171-173), so it runs as ONE fused launch over 2 N sequences (480 workgroups at B = 8: two per CU) instead of
        two launches of 240; a 'cross' layer (:174-176, the second call consumes the first call's output) writes its two results into the
        halves of the next matrix, so no copy separates the layers."""
        R = N * L
        F = engine.tap("tf.in", engine.rows_cat(f0, f1))
        for li, (layer, name) in enumerate(zip(self.layers, self.layer_names)):
            if name == 'self':
                F = engine.tap("tf.layer%d" % li, engine.loftr_layer(F, F, layer, 2 * N, L, L))
            elif name == 'cross':
                a, b = engine.rows_split(F, R)
                G = torch.empty_like(F)
                cg = engine.CrossGrad()       # the pair's backward writes d [a; b] in place (no gradient-add passes, no row copies)
                a2 = engine.loftr_layer(a, b, layer, N, L, L, out=G[:R], cross=cg, cross_role=1)
                b2 = engine.loftr_layer(b, a2, layer, N, L, L, out=G[R:], cross=cg, cross_role=2)
                F = engine.tap("tf.layer%d" % li, engine.rows_join(a2, b2, G))
            else:
                raise KeyError
        return engine.rows_split(F, R)

    def forward(self, feat0, feat1, mask0=None, mask1=None):
        assert self.d_model == feat0.size(2), "the feature number of src and transformer must be equal"
        if mask0 is not None or mask1 is not None:
            raise NotImplementedError('masks are always None on the RIDERS path')
        N, L, C = feat0.shape
        S = feat1.shape[1]

        def run(a, b):
            f0, f1 = self._fwd(engine.tokens_in(a, C), engine.tokens_in(b, C), N, L, S)
            return (engine.alias_cast_view(f0, a.dtype, (N, L, C)), engine.alias_cast_view(f1, b.dtype, (N, S, C)))
        return engine.run_region(run, (feat0, feat1), list(self.parameters()))
