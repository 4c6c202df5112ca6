"""torch.autograd.Function wrappers of the per-operator entry points of libpsg (include/psg.h "Per-operator entry points"):
what the STAND-ALONE forwards of the reference's public modules run on.

    SetAbstractionMLP      PointNet/models/pointnet_util.py:126-140, 200-205   (grouping + shared MLP + max over K)
    FeaturePropagationMLP  PointNet/models/pointnet_util.py:301-319            (3-NN interpolation + concat + MLP)
    EdgeConv               ResGCN/gcn_lib/dense/torch_vertex.py:31-35          (EdgeConv2d with BasicConv's Conv->ReLU->BN)
    MRConv                 ResGCN/gcn_lib/dense/torch_vertex.py:8-20           (max-relative convolution)
    pairwise_distance      ResGCN/gcn_lib/dense/torch_edge.py:32-42

Every tensor is a CUDA tensor; PyTorch supplies device memory, the current stream and the autograd graph, the arithmetic is
in the HIP kernels.  Eval-mode BatchNorm is folded here (double precision, rounded once, like runtime.fold_state_dict).
Gradients are produced for the input features only (the attack path differentiates w.r.t. its input; parameters and
coordinates receive none), and there is no CPU fallback."""
import ctypes

import torch

from pointsecguard_amd import _lib, runtime

BN_EPS = 1e-5


def fold_conv_bn(conv, bn):
    """Conv(1x1) followed by eval BatchNorm -> (W' [cout][cin], b' [cout]) float32 on the conv's device."""
    w = conv.weight.detach().double().reshape(conv.weight.shape[0], -1)
    b = conv.bias.detach().double() if conv.bias is not None else torch.zeros(w.shape[0], dtype=torch.float64, device=w.device)
    s = bn.weight.detach().double() / torch.sqrt(bn.running_var.detach().double() + bn.eps)
    return (w * s[:, None]).float().contiguous(), ((b - bn.running_mean.detach().double()) * s + bn.bias.detach().double()).float().contiguous()


def bn_affine(bn):
    s = bn.weight.detach().double() / torch.sqrt(bn.running_var.detach().double() + bn.eps)
    return s.float().contiguous(), (bn.bias.detach().double() - bn.running_mean.detach().double() * s).float().contiguous()


def _ptr_array(tensors):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() if t is not None else None for t in tensors])


def _int_array(vals):
    return (ctypes.c_int * len(vals))(*vals)


def _layers(ws, bs, dev):
    ws = [w.to(dev).float().contiguous() for w in ws]
    bs = [b.to(dev).float().contiguous() for b in bs]
    return ws, bs, [w.t().contiguous() for w in ws]


class SetAbstractionMLP(torch.autograd.Function):
    """(xyz [B,N,3], feat [B,N,D] or None, new_xyz [B,S,3], gidx [B,S,K] int32, layers) -> pooled [B,S,C_last]."""

    @staticmethod
    def forward(ctx, feat, xyz, new_xyz, gidx, ws, bs, feat_first):
        runtime.require_cuda(xyz, "xyz", torch.float32)
        B, N, _ = xyz.shape
        S, K = gidx.shape[1], gidx.shape[2]
        D = 0 if feat is None else feat.shape[2]
        dev = xyz.device
        ws, bs, wts = _layers(ws, bs, dev)
        widths = [w.shape[0] for w in ws]
        rows = B * S * K
        st = runtime.stream
        grouped = torch.empty(rows, D + 3, device=dev)
        _lib.call("psg_group_rows", runtime.ptr(xyz), runtime.ptr(feat), runtime.ptr(new_xyz), runtime.ptr(gidx), B, N, S, K, D,
                  1 if feat_first else 0, runtime.ptr(grouped), st())
        wmax = max(widths + [D + 3])
        sa, sb = torch.empty(rows, wmax, device=dev), torch.empty(rows, wmax, device=dev)
        masks = [torch.empty(rows, (w + 31) // 32, dtype=torch.int32, device=dev) for w in widths]
        out = torch.empty(B, S, widths[-1], device=dev)
        arg = torch.empty(B, S, widths[-1], dtype=torch.uint8, device=dev)
        _lib.call("psg_sa_mlp_max_fwd", runtime.ptr(grouped), B * S, K, D + 3, len(ws), _int_array(widths), _ptr_array(ws),
                  _ptr_array(bs), runtime.ptr(sa), runtime.ptr(sb), _ptr_array(masks), runtime.ptr(out), runtime.ptr(arg), st())
        ctx.save_for_backward(gidx, arg, *masks, *wts)
        ctx.meta = (B, N, S, K, D, widths, bool(feat_first), sa, sb)
        return out

    @staticmethod
    def backward(ctx, dout):
        B, N, S, K, D, widths, feat_first, sa, sb = ctx.meta
        if D == 0:
            return (None,) * 7
        saved = ctx.saved_tensors
        gidx, arg = saved[0], saved[1]
        n = len(widths)
        masks, wts = list(saved[2:2 + n]), list(saved[2 + n:2 + 2 * n])
        dev = dout.device
        st = runtime.stream
        dout = dout.contiguous().float()
        drows = torch.empty(B * S * K, D + 3, device=dev)
        _lib.call("psg_sa_mlp_max_bwd", runtime.ptr(dout), runtime.ptr(arg), B * S, K, D + 3, n, _int_array(widths), _ptr_array(wts),
                  _ptr_array(masks), runtime.ptr(sa), runtime.ptr(sb), runtime.ptr(drows), st())
        dfeat = torch.empty(B, N, D, device=dev)
        _lib.call("psg_group_rows_bwd", runtime.ptr(drows), runtime.ptr(gidx), B, N, S, K, D, 1 if feat_first else 0,
                  runtime.ptr(dfeat), st())
        return dfeat, None, None, None, None, None, None


class FeaturePropagationMLP(torch.autograd.Function):
    """(feat1 [B,N,D1] or None, feat2 [B,S,D2], idx [B,N,3] int32, w [B,N,3], layers) -> [B,N,C_last]."""

    @staticmethod
    def forward(ctx, feat1, feat2, idx, w, ws, bs):
        runtime.require_cuda(feat2, "points2", torch.float32)
        B, S, D2 = feat2.shape
        N = idx.shape[1]
        D1 = 0 if feat1 is None else feat1.shape[2]
        dev = feat2.device
        ws, bs, wts = _layers(ws, bs, dev)
        widths = [x.shape[0] for x in ws]
        st = runtime.stream
        cur = torch.empty(B * N, D1 + D2, device=dev)
        _lib.call("psg_three_interp_fwd", runtime.ptr(feat2), runtime.ptr(idx), runtime.ptr(w), runtime.ptr(feat1), B, N, S, D1, D2,
                  runtime.ptr(cur), st())
        masks, c = [], D1 + D2
        for wl, bl in zip(ws, bs):
            nxt = torch.empty(B * N, wl.shape[0], device=dev)
            mk = torch.empty(B * N, (wl.shape[0] + 31) // 32, dtype=torch.int32, device=dev)
            _lib.call("psg_pw_mlp_fwd", runtime.ptr(cur), c, B * N, c, runtime.ptr(wl), runtime.ptr(bl), 1, wl.shape[0],
                      runtime.ptr(nxt), wl.shape[0], runtime.ptr(mk), None, None, st())
            masks.append(mk)
            cur, c = nxt, wl.shape[0]
        ctx.save_for_backward(idx, w, *masks, *wts)
        ctx.meta = (B, N, S, D1, D2, widths)
        return cur.view(B, N, c)

    @staticmethod
    def backward(ctx, dout):
        B, N, S, D1, D2, widths = ctx.meta
        saved = ctx.saved_tensors
        idx, w = saved[0], saved[1]
        n = len(widths)
        masks, wts = list(saved[2:2 + n]), list(saved[2 + n:2 + 2 * n])
        dev = dout.device
        st = runtime.stream
        # through the last layer's own ReLU first (its bits), then layer by layer
        g = dout.contiguous().float().view(B * N, widths[-1]).clone()
        _lib.call("psg_apply_relu_bits", runtime.ptr(g), widths[-1], runtime.ptr(masks[-1]), None, B * N, widths[-1], st())
        for l in range(n - 1, -1, -1):
            c_in = widths[l - 1] if l else D1 + D2
            nxt = torch.empty(B * N, c_in, device=dev)
            _lib.call("psg_pw_mlp_bwd", runtime.ptr(g), widths[l], B * N, widths[l], runtime.ptr(wts[l]),
                      runtime.ptr(masks[l - 1]) if l else None, c_in, runtime.ptr(nxt), c_in, st())
            g = nxt
        d1 = g[:, :D1].reshape(B, N, D1).contiguous() if D1 else None
        d2 = torch.empty(B, S, D2, device=dev)
        _lib.call("psg_three_interp_bwd", runtime.ptr(g), D1 + D2, D1, runtime.ptr(idx), runtime.ptr(w), B, N, S, D2, runtime.ptr(d2), st())
        return d1, d2, None, None, None, None


class EdgeConv(torch.autograd.Function):
    """(x [B,N,C] point-major, nbr [B,N,16] int32 room-local, W [64][2C], b [64], scale, shift) -> [B,N,64]."""

    @staticmethod
    def forward(ctx, x, nbr, weight, bias, scale, shift):
        runtime.require_cuda(x, "x", torch.float32)
        B, N, C = x.shape
        if weight.shape[0] != 64 or nbr.shape[2] != 16:
            raise NotImplementedError("the gfx950 EdgeConv kernels are specialised for 64 output channels and k = 16 neighbours "
                                      "(the reference's configuration, ResGCN/sem_seg_dense/config.py:52-56)")
        dev = x.device
        w = weight.detach().to(dev).double().reshape(64, 2 * C)
        w1, w2 = w[:, :C], w[:, C:]
        wcat = torch.cat([w1 - w2, w2], 0).float().contiguous()                    # W.[x_i, x_j - x_i] = (W1 - W2) x_i + W2 x_j
        bcat = torch.cat([bias.detach().to(dev).float(), torch.zeros(64, device=dev)]).contiguous()
        R = B * N
        pq = torch.empty(R, 128, device=dev)
        out = torch.empty(B, N, 64, device=dev)
        arg = torch.empty(R, 64, dtype=torch.uint8, device=dev)
        scale, shift = scale.to(dev).float().contiguous(), shift.to(dev).float().contiguous()
        _lib.call("psg_edgeconv_fwd", runtime.ptr(x), C, R, N, C, runtime.ptr(nbr), runtime.ptr(wcat), runtime.ptr(bcat),
                  runtime.ptr(scale), runtime.ptr(shift), runtime.ptr(pq), runtime.ptr(out), 64, runtime.ptr(arg), runtime.stream())
        ctx.save_for_backward(nbr, arg, scale, wcat.t().contiguous())
        ctx.meta = (B, N, C, pq)
        return out

    @staticmethod
    def backward(ctx, dy):
        nbr, arg, scale, wcat_t = ctx.saved_tensors
        B, N, C, dpq = ctx.meta
        dy = dy.contiguous().float()
        dx = torch.empty(B, N, C, device=dy.device)
        _lib.call("psg_edgeconv_bwd", runtime.ptr(dy), 64, B * N, N, C, runtime.ptr(nbr), runtime.ptr(arg), runtime.ptr(scale),
                  runtime.ptr(wcat_t), runtime.ptr(dpq), runtime.ptr(dx), C, runtime.stream())
        return dx, None, None, None, None, None


class MRConv(torch.autograd.Function):
    """(x [B,N,C], nbr [B,N,16] int32, W [M][2C], b [M], scale, shift) -> [B,N,M]: BasicConv(cat[x_i, max_k (x_j - x_i)])
    with Conv -> ReLU -> BatchNorm (torch_vertex.py:8-20, torch_nn.py:55-75)."""

    @staticmethod
    def forward(ctx, x, nbr, weight, bias, scale, shift):
        runtime.require_cuda(x, "x", torch.float32)
        B, N, C = x.shape
        if nbr.shape[2] != 16:
            raise NotImplementedError("the gfx950 graph-convolution kernels are specialised for k = 16 neighbours")
        dev = x.device
        w = weight.detach().to(dev).float().reshape(weight.shape[0], 2 * C).contiguous()
        M, R = w.shape[0], B * N
        st = runtime.stream
        cat = torch.empty(R, 2 * C, device=dev)
        arg = torch.empty(R, C, dtype=torch.uint8, device=dev)
        _lib.call("psg_mrconv_gather_fwd", runtime.ptr(x), C, R, N, C, runtime.ptr(nbr), runtime.ptr(cat), runtime.ptr(arg), st())
        out = torch.empty(B, N, M, device=dev)
        bits = torch.empty(R, (M + 31) // 32, dtype=torch.int32, device=dev)
        scale, shift = scale.to(dev).float().contiguous(), shift.to(dev).float().contiguous()
        b = bias.detach().to(dev).float().contiguous()
        _lib.call("psg_pw_mlp_fwd", runtime.ptr(cat), 2 * C, R, 2 * C, runtime.ptr(w), runtime.ptr(b), 1, M, runtime.ptr(out), M,
                  runtime.ptr(bits), runtime.ptr(scale), runtime.ptr(shift), st())
        ctx.save_for_backward(nbr, arg, bits, scale, w.t().contiguous())
        ctx.meta = (B, N, C, M)
        return out

    @staticmethod
    def backward(ctx, dy):
        nbr, arg, bits, scale, wt = ctx.saved_tensors
        B, N, C, M = ctx.meta
        R = B * N
        st = runtime.stream
        g = dy.contiguous().float().view(R, M).clone()
        _lib.call("psg_apply_relu_bits", runtime.ptr(g), M, runtime.ptr(bits), runtime.ptr(scale), R, M, st())
        dcat = torch.empty(R, 2 * C, device=g.device)
        _lib.call("psg_pw_mlp_bwd", runtime.ptr(g), M, R, M, runtime.ptr(wt), None, 2 * C, runtime.ptr(dcat), 2 * C, st())
        dx = torch.empty(B, N, C, device=g.device)
        _lib.call("psg_mrconv_gather_bwd", runtime.ptr(dcat), R, N, C, runtime.ptr(nbr), runtime.ptr(arg), runtime.ptr(dx), C, st())
        return dx, None, None, None, None, None


def pairwise_distance(x):
    """x [B,N,C] -> [B,N,N] fp32 with the reference's evaluation order (torch_edge.py:32-42); no gradient (the reference
    calls it under no_grad on detached features, torch_edge.py:50-53)."""
    x = runtime.require_cuda(x.detach().float().contiguous(), "x", torch.float32)
    B, N, C = x.shape
    sq = torch.empty(B * N, device=x.device)
    out = torch.empty(B, N, N, device=x.device)
    _lib.call("psg_gcn_pairwise_distance", runtime.ptr(x), B, N, C, runtime.ptr(sq), runtime.ptr(out), runtime.stream())
    return out


def global_max(x):
    """torch.max_pool2d(x, [N, 1]) of DenseDeepGCN.forward (architecture.py:64) for point-major x [B, N, C] (C a multiple
    of 64): (max [B, C], arg-max row [B, C] int32, lowest row on equal values)."""
    runtime.require_cuda(x, "x", torch.float32)
    B, N, C = x.shape
    scratch = torch.empty(B * C, dtype=torch.int64, device=x.device)
    mx = torch.empty(B, C, dtype=torch.float32, device=x.device)
    arg = torch.empty(B, C, dtype=torch.int32, device=x.device)
    _lib.call("psg_global_max", runtime.ptr(x.contiguous()), B, N, C, runtime.ptr(scratch), runtime.ptr(mx), runtime.ptr(arg),
              runtime.stream())
    return mx, arg
