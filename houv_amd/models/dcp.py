"""Mirror of registration/models/dcp.py (`Model`, :384-430): the DCP feature head -- DGCNN embedding (:269-301),
one-layer encoder/decoder Transformer pointer (:69-243, :304-328), soft correspondences + SVD head (:331-381) -- as an
INFERENCE pipeline over the gfx950 kernels of include/houv_hip.h (houv_knn, houv_edgeconv1, houv_gemm_f32 on fp32 MFMA,
houv_max_over_k, houv_layernorm, houv_softmax_rows, houv_softmax_corr, houv_kabsch).  PyTorch only owns the buffers.

The module tree and parameter names equal the reference's, so a reference checkpoint's ``net_state_dict`` loads with
``load_state_dict`` unchanged (the repository itself ships no trained weights: tests use seeded random ones).
BatchNorm is applied in eval mode (running statistics), as the reference's test drivers do (test.py:47)."""
import math

import torch
import torch.nn as nn

from .. import ops
from ..train_utils import rmse_loss, rotation_error, translation_error

K_NEIGHBOURS = 20          # get_graph_feature(x, k=20), dcp.py:44
N_HEADS, D_MODEL, D_FF = 4, 512, 1024


class _Linear(nn.Module):
    def __init__(self, n_in, n_out):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(n_out, n_in).uniform_(-1, 1) / math.sqrt(n_in))
        self.bias = nn.Parameter(torch.zeros(n_out))


class _Conv1x1(nn.Module):
    def __init__(self, n_in, n_out):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(n_out, n_in, 1, 1).uniform_(-1, 1) / math.sqrt(n_in))


class _BN(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self.eps = 1e-5

    def folded(self):
        """Eval-mode BatchNorm as a per-channel (scale, shift) for the GEMM epilogue.  Cached: the fold used to be recomputed by
        five tiny torch kernels per layer and call (~600 launches per bench step, 4 % of it); any in-place change of the four
        tensors (load_state_dict, .to(), an optimiser step) bumps their version counters and invalidates the cache."""
        key = (self.weight._version, self.bias._version, self.running_mean._version, self.running_var._version,
               self.weight.device, self.weight.data_ptr())
        if getattr(self, "_folded_key", None) != key:
            with torch.no_grad():
                scale = self.weight / torch.sqrt(self.running_var + self.eps)
                self._folded = (scale.contiguous(), (self.bias - self.running_mean * scale).contiguous())
            self._folded_key = key
        return self._folded


class _LayerNorm(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.a_2 = nn.Parameter(torch.ones(d))
        self.b_2 = nn.Parameter(torch.zeros(d))
        self.eps = 1e-6

    def forward(self, x, residual=None):
        return ops.layernorm(x, self.a_2, self.b_2, self.eps, residual)


class _Sublayer(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.norm = _LayerNorm(d)


class _Attention(nn.Module):
    """MultiHeadedAttention (dcp.py:198-229): 4 heads x 128; `linears` = (Wq, Wk, Wv, Wo)."""

    def __init__(self):
        super().__init__()
        self.linears = nn.ModuleList([_Linear(D_MODEL, D_MODEL) for _ in range(4)])

    def forward(self, q_in, kv_in, residual):
        P, Nq, _ = q_in.shape
        Nk = kv_in.shape[1]
        dk = D_MODEL // N_HEADS
        lq, lk, lv, lo = self.linears
        Q = ops.gemm(q_in.reshape(P * Nq, D_MODEL), lq.weight, shift=lq.bias).view(P, Nq, N_HEADS, dk)
        Kt = ops.gemm(kv_in.reshape(P * Nk, D_MODEL), lk.weight, shift=lk.bias).view(P, Nk, N_HEADS, dk)
        V = ops.gemm(kv_in.reshape(P * Nk, D_MODEL), lv.weight, shift=lv.bias).view(P, Nk, N_HEADS, dk)
        if ops.FUSED_ATTENTION:
            # softmax(Q K^T / sqrt(dk)) V per head in one kernel: the [P,4,Nq,Nk] scores never reach HBM
            ctx = ops.attention(Q, Kt, V, 1.0 / math.sqrt(dk))
        else:
            # scores[p,h] = Q_ph K_ph^T / sqrt(dk): per-head operands are strided views (row stride 512), no copies
            scores = ops.gemm(Q.permute(0, 2, 1, 3), Kt.permute(0, 2, 1, 3), trans_b=True, alpha=1.0 / math.sqrt(dk))
            ops.softmax_rows_(scores)
            ctx = torch.empty((P, Nq, N_HEADS, dk), dtype=torch.float32, device=q_in.device)
            ops.gemm(scores, V.permute(0, 2, 1, 3), ctx.permute(0, 2, 1, 3), trans_b=False)
        out = ops.gemm(ctx.view(P * Nq, D_MODEL), lo.weight, shift=lo.bias, residual=residual.reshape(P * Nq, D_MODEL))
        return out.view(P, Nq, D_MODEL)


class _FeedForward(nn.Module):
    def __init__(self):
        super().__init__()
        self.w_1 = _Linear(D_MODEL, D_FF)
        self.w_2 = _Linear(D_FF, D_MODEL)

    def forward(self, x, residual):
        P, N, _ = x.shape
        h = ops.gemm(x.reshape(P * N, D_MODEL), self.w_1.weight, shift=self.w_1.bias, relu=True)
        return ops.gemm(h, self.w_2.weight, shift=self.w_2.bias, residual=residual.reshape(P * N, D_MODEL)).view(P, N, D_MODEL)


class _EncoderLayer(nn.Module):
    def __init__(self):
        super().__init__()
        self.self_attn = _Attention()
        self.feed_forward = _FeedForward()
        self.sublayer = nn.ModuleList([_Sublayer(D_MODEL) for _ in range(2)])

    def forward(self, x):
        n = self.sublayer[0].norm(x)
        x = self.self_attn(n, n, residual=x)                          # x + attn(norm(x))   (dcp.py:162-163,175)
        return self.feed_forward(self.sublayer[1].norm(x), residual=x)


class _DecoderLayer(nn.Module):
    def __init__(self):
        super().__init__()
        self.self_attn = _Attention()
        self.src_attn = _Attention()
        self.feed_forward = _FeedForward()
        self.sublayer = nn.ModuleList([_Sublayer(D_MODEL) for _ in range(3)])

    def forward(self, x, memory):
        n = self.sublayer[0].norm(x)
        x = self.self_attn(n, n, residual=x)
        x = self.src_attn(self.sublayer[1].norm(x), memory, residual=x)       # dcp.py:194
        return self.feed_forward(self.sublayer[2].norm(x), residual=x)


class _Stack(nn.Module):
    def __init__(self, layer):
        super().__init__()
        self.layers = nn.ModuleList([layer])
        self.norm = _LayerNorm(D_MODEL)


class _EncoderDecoder(nn.Module):
    def __init__(self):
        super().__init__()
        self.encoder = _Stack(_EncoderLayer())
        self.decoder = _Stack(_DecoderLayer())

    def forward(self, src, tgt, add_to):
        """decode(encode(src), tgt) (dcp.py:83-93) with the caller's `tgt_embedding + pointer output` (:405-406)
        fused into the final LayerNorm as a residual."""
        memory = self.encoder.norm(self.encoder.layers[0](src))
        return self.decoder.norm(self.decoder.layers[0](tgt, memory), residual=add_to)


class Transformer(nn.Module):
    def __init__(self, args=None):
        super().__init__()
        self.model = _EncoderDecoder()


class DGCNN(nn.Module):
    def __init__(self, emb_dims=512):
        super().__init__()
        self.conv1, self.conv2 = _Conv1x1(6, 64), _Conv1x1(64, 64)
        self.conv3, self.conv4 = _Conv1x1(64, 128), _Conv1x1(128, 256)
        self.conv5 = _Conv1x1(512, emb_dims)
        self.bn1, self.bn2, self.bn3, self.bn4, self.bn5 = _BN(64), _BN(64), _BN(128), _BN(256), _BN(emb_dims)

    def forward(self, xyz):
        """xyz[P,N,3] -> embedding[P,N,512] (token-major; the reference returns [P,512,N])."""
        P, N, _ = xyz.shape
        k = K_NEIGHBOURS
        idx = ops.knn(xyz, k)
        cat = torch.empty((P * N, 512), dtype=torch.float32, device=xyz.device)
        s, h = self.bn1.folded()
        act = ops.edgeconv1(xyz, idx, self.conv1.weight.reshape(64, 6).contiguous(), s, h)
        ops.max_over_k(act, k, cat, 0)
        col = 64
        for conv, bn in ((self.conv2, self.bn2), (self.conv3, self.bn3), (self.conv4, self.bn4)):
            s, h = bn.folded()
            w = conv.weight.reshape(conv.weight.shape[0], conv.weight.shape[1])
            act = ops.gemm(act, w, scale=s, shift=h, relu=True)
            ops.max_over_k(act, k, cat, col)
            col += w.shape[0]
        s, h = self.bn5.folded()
        emb = ops.gemm(cat, self.conv5.weight.reshape(512, 512), scale=s, shift=h, relu=True)
        return emb.view(P, N, 512)


class _Conv1d1x1(nn.Module):
    def __init__(self, n_in, n_out):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(n_out, n_in, 1).uniform_(-1, 1) / math.sqrt(n_in))


class PointNet(nn.Module):
    """The per-point MLP embedding of dcp.py:246-266 (3->64->64->64->128->512, BatchNorm1d + ReLU after each 1x1 conv):
    five MFMA GEMMs with the folded-BN/ReLU epilogue.  `Model` uses DGCNN, like the reference (:388)."""

    def __init__(self, emb_dims=512):
        super().__init__()
        dims = (3, 64, 64, 64, 128, emb_dims)
        for i in range(5):
            setattr(self, f"conv{i + 1}", _Conv1d1x1(dims[i], dims[i + 1]))
            setattr(self, f"bn{i + 1}", _BN(dims[i + 1]))

    def forward(self, xyz):
        """xyz[P,N,3] -> [P,N,512] token-major (the reference takes [P,3,N] and returns [P,512,N])."""
        P, N, _ = xyz.shape
        x = xyz.reshape(P * N, 3).contiguous()
        for i in range(1, 6):
            conv, bn = getattr(self, f"conv{i}"), getattr(self, f"bn{i}")
            s, h = bn.folded()
            x = ops.gemm(x, conv.weight.reshape(conv.weight.shape[0], conv.weight.shape[1]), scale=s, shift=h, relu=True)
        return x.view(P, N, -1)


class SVDHead(nn.Module):
    def __init__(self, args=None):
        super().__init__()
        self.emb_dims = 512
        self.reflect = nn.Parameter(torch.eye(3), requires_grad=False)
        self.reflect[2, 2] = -1

    def forward(self, src_embedding, tgt_embedding, src, tgt):
        """embeddings token-major [P,N,512]/[P,M,512]; src[P,N,3], tgt[P,M,3] -> R[P,3,3], t[P,3] (dcp.py:338-381)."""
        scores = ops.gemm(src_embedding, tgt_embedding, trans_b=True, alpha=1.0 / math.sqrt(self.emb_dims))
        src_corr = ops.softmax_corr(scores, tgt)                       # [P,3,N]
        return ops.kabsch(src.transpose(1, 2).contiguous(), src_corr)


class Model(nn.Module):
    """``Model(args).forward(src[B,N,3], tgt[B,M,3], T_gt=None, prefix="train")`` (dcp.py:384-430): returns T_12[B,4,4],
    or (loss, r_err, t_err, rmse, rt_mse) when T_gt is given.  ``pairs_per_chunk`` bounds the workspace (the attention
    scores of one pair are 4 x N x M floats)."""

    def __init__(self, args=None, pairs_per_chunk=8):
        super().__init__()
        self.emb_dims = 512
        self.cycle = False
        self.emb_nn = DGCNN(emb_dims=self.emb_dims)
        self.pointer = Transformer(args=args)
        self.head = SVDHead(args=args)
        self.pairs_per_chunk = pairs_per_chunk
        self.use_graphs = True                 # replay a chunk's launches as one HIP graph from its third occurrence on (_graphed_chunk)
        self._graphs = {}

    def _chunk(self, s, t):
        es, et = self.emb_nn(s), self.emb_nn(t)
        ed = self.pointer.model
        tgt_e = ed(es, et, add_to=et)          # tgt_embedding + model(src, tgt)   (dcp.py:325,406)
        src_e = ed(et, es, add_to=es)          # src_embedding + model(tgt, src)   (dcp.py:326,405)
        return self.head(src_e, tgt_e, s, t)

    def _state_signature(self):
        """Identity + version of every parameter and buffer: a captured graph bakes their addresses (and the cached BatchNorm folds)."""
        return tuple((t.data_ptr(), t._version) for t in list(self.parameters()) + list(self.buffers()))

    def _graphed_chunk(self, s, t, sig):
        """A chunk's ~120 launches replayed as ONE HIP graph (round 3: the launches' host side was 10 % of a bench step).  Per
        (chunk shape, device): the first call runs eagerly (it is also the warm-up a capture needs), the second captures with static
        input buffers, later ones copy the inputs in and replay.  The library's stream-ordered workspace (houv_attention_f32) is
        captured as allocation nodes; any failure falls back to eager execution for good."""
        key = (tuple(s.shape), tuple(t.shape), s.device)
        ent = self._graphs.get(key)
        if ent is None or ent["sig"] != sig:
            self._graphs[key] = {"sig": sig, "graph": None}
            return self._chunk(s, t)
        if ent["graph"] is None:
            try:
                ent["s"], ent["t"] = s.clone(), t.clone()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    ent["out"] = self._chunk(ent["s"], ent["t"])
                ent["graph"] = g
            except Exception:                                  # capture is an optimisation, never a requirement
                self.use_graphs = False
                self._graphs.clear()
                torch.cuda.synchronize(s.device)
                return self._chunk(s, t)
        ent["s"].copy_(s)
        ent["t"].copy_(t)
        ent["graph"].replay()
        return tuple(o.clone() for o in ent["out"])

    @torch.no_grad()
    def registration(self, src, tgt):
        from .. import ops
        Rs, ts = [], []
        graphs = self.use_graphs and src.is_cuda and ops.GEMM_LOG is None and not torch.cuda.is_current_stream_capturing()
        sig = self._state_signature() if graphs else None
        for s0 in range(0, src.shape[0], self.pairs_per_chunk):
            s = src[s0:s0 + self.pairs_per_chunk].contiguous().float()
            t = tgt[s0:s0 + self.pairs_per_chunk].contiguous().float()
            R, tr = self._graphed_chunk(s, t, sig) if graphs else self._chunk(s, t)
            Rs.append(R)
            ts.append(tr)
        return torch.cat(Rs, 0), torch.cat(ts, 0)

    def forward(self, src, tgt, T_gt=None, prefix="train"):
        R, t = self.registration(src, tgt)
        B = R.shape[0]
        T_12 = torch.zeros((B, 4, 4), dtype=torch.float32, device=R.device)      # rt_to_transformation (train_utils.py:75-78)
        T_12[:, :3, :3] = R
        T_12[:, :3, 3] = t
        T_12[:, 3, 3] = 1.0
        if T_gt is None:
            return T_12
        r_err = rotation_error(T_12[:, :3, :3], T_gt[:, :3, :3])
        t_err = translation_error(T_12[:, :3, 3], T_gt[:, :3, 3])
        rmse = rmse_loss(src, T_12, T_gt)
        eye = torch.eye(4, device=T_gt.device).expand_as(T_gt)
        loss = torch.nn.functional.mse_loss(T_12 @ torch.inverse(T_gt), eye)
        cos = torch.clamp((torch.einsum('bii->b', T_12[:, :3, :3] @ T_gt[:, :3, :3].transpose(1, 2)) - 1) / 2, -1, 1)
        rt_mse = torch.acos(cos) + t_err                                           # rotation_geodesic_error (train_utils.py:98-110)
        return loss, r_err, t_err, rmse, rt_mse
