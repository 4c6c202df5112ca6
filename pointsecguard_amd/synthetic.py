"""Synthetic S3DIS-shaped rooms (no dataset ships with the reference or this repo).

A "room" is one 4096-point x 9-channel block with the channel contract produced by the
reference's whole-scene loader (PointNet/data_utils/S3DISDataLoader.py:155-171):
channels 0:3 block-centred xyz, 3:6 rgb in [0,1], 6:9 room-normalised xyz in [0,1].
The reference harness feeds blocks channel-major, [B,9,N]
(PointNet/NB_nontarget_test_semseg.py:163-165).
"""
import numpy as np
import torch

NUM_POINT = 4096
NUM_CLASSES = 13


def make_rooms(batch, seed, num_point=NUM_POINT, structured=False):
    """Return float32 [batch, num_point, 9] (point-major) rooms, seeded and platform independent."""
    g = torch.Generator().manual_seed(int(seed))
    if not structured:
        u = torch.rand(batch, num_point, 9, generator=g, dtype=torch.float32)
        rooms = u.clone()
        rooms[..., 0] = u[..., 0] - 0.5
        rooms[..., 1] = u[..., 1] - 0.5
        rooms[..., 2] = u[..., 2] * 3.0
        return rooms.numpy()
    # structured rooms: points on floor / ceiling / 4 walls / a box, colour correlated with surface
    u = torch.rand(batch, num_point, 6, generator=g, dtype=torch.float32)
    surf = torch.randint(0, 7, (batch, num_point), generator=g)
    x = u[..., 0] - 0.5
    y = u[..., 1] - 0.5
    z = u[..., 2] * 3.0
    jit = (u[..., 3] - 0.5) * 0.01
    x = torch.where(surf == 2, -0.5 + jit.abs(), x)
    x = torch.where(surf == 3, 0.5 - jit.abs(), x)
    y = torch.where(surf == 4, -0.5 + jit.abs(), y)
    y = torch.where(surf == 5, 0.5 - jit.abs(), y)
    z = torch.where(surf == 0, jit.abs(), z)
    z = torch.where(surf == 1, 3.0 - jit.abs(), z)
    box = surf == 6
    x = torch.where(box, x * 0.4, x)
    y = torch.where(box, y * 0.4, y)
    z = torch.where(box, 0.8 + jit, z)
    base = torch.tensor([[.7, .6, .5], [.9, .9, .9], [.8, .3, .3], [.3, .8, .3],
                         [.3, .3, .8], [.8, .8, .3], [.5, .3, .1]], dtype=torch.float32)
    rgb = (base[surf] + (u[..., 3:6] - 0.5) * 0.2).clamp(0, 1)
    rooms = torch.empty(batch, num_point, 9, dtype=torch.float32)
    rooms[..., 0], rooms[..., 1], rooms[..., 2] = x, y, z
    rooms[..., 3:6] = rgb
    rooms[..., 6] = x + 0.5
    rooms[..., 7] = y + 0.5
    rooms[..., 8] = z / 3.0
    return rooms.numpy()



def make_rooms_with_duplicates(batch, seed):
    """Rooms that contain EXACT duplicates of points, the way the reference's loader makes them (S3DISDataLoader.py:149-154
    samples a block's points with replacement): every room keeps 3072 distinct points of make_rooms(batch, seed) and fills the
    other 1024 slots with copies (whole 9-channel rows) drawn with replacement, then is shuffled."""
    room = make_rooms(batch, seed)
    rng = np.random.default_rng(seed + 1000)
    for b in range(batch):
        keep = rng.permutation(4096)[:3072]
        idx = np.concatenate([keep, rng.choice(keep, 1024, replace=True)])
        room[b] = room[b][rng.permutation(idx)]
    return room

def rule_labels(rooms):
    """Deterministic 13-class labels for synthetic rooms: 3*floor(4z/3) + argmax(rgb), class 12 where x*y > 0.15.

    Gives non-degenerate accuracy / mIoU / attack-success numbers with the fitted fixture weights
    (tests/golden/pn2_weights.npz). rooms: [B,N,9] -> int64 [B,N].
    """
    rooms = np.asarray(rooms)
    x, y, z = rooms[..., 0], rooms[..., 1], rooms[..., 2]
    band = np.clip(np.floor(z * np.float32(4.0 / 3.0)).astype(np.int64), 0, 3)
    lab = 3 * band + np.argmax(rooms[..., 3:6], axis=-1)
    lab = np.where(x * y > np.float32(0.15), 12, lab)
    return lab.astype(np.int64)


# (in_channel, mlp_list) of sa1..sa4 and (in_channel, mlp) of fp4..fp1 of the MSG network
# (PointNet/models/pointnet2_sem_seg_msg.py:10-17 of the reference)
MSG_SA = ((9, ((16, 16, 32), (32, 32, 64))), (96, ((64, 64, 128), (64, 96, 128))),
          (256, ((128, 196, 256), (128, 196, 256))), (512, ((256, 256, 512), (256, 384, 512))))
MSG_FP = (("fp4", 1536, (256, 256)), ("fp3", 512, (256, 256)), ("fp2", 352, (256, 128)), ("fp1", 128, (128, 128, 128)))


def msg_state_dict(seed):
    """Seeded random weights for pointnet2_sem_seg_msg.get_model(13) with the reference's state_dict keys and
    shapes: He-scaled conv weights and non-trivial eval BatchNorm statistics, so that activations, ReLU patterns
    and gradients are non-degenerate (no checkpoint of this variant ships with the reference)."""
    rng = np.random.RandomState(int(seed))
    sd = {}

    def conv(name, cin, cout, dims):
        shape = (cout, cin) + (1,) * dims
        sd[name + ".weight"] = (rng.standard_normal(shape) * np.sqrt(2.0 / cin)).astype(np.float32)
        sd[name + ".bias"] = rng.uniform(-0.1, 0.1, cout).astype(np.float32)

    def bn(name, c):
        sd[name + ".weight"] = rng.uniform(0.8, 1.2, c).astype(np.float32)
        sd[name + ".bias"] = rng.uniform(-0.1, 0.1, c).astype(np.float32)
        sd[name + ".running_mean"] = rng.uniform(-0.1, 0.1, c).astype(np.float32)
        sd[name + ".running_var"] = rng.uniform(0.5, 1.5, c).astype(np.float32)
        sd[name + ".num_batches_tracked"] = np.array(1, np.int64)

    for l, (cin, mlps) in enumerate(MSG_SA, start=1):
        for i, mlp in enumerate(mlps):
            last = cin + 3
            for j, c in enumerate(mlp):
                conv("sa%d.conv_blocks.%d.%d" % (l, i, j), last, c, 2)
                bn("sa%d.bn_blocks.%d.%d" % (l, i, j), c)
                last = c
    for name, cin, mlp in MSG_FP:
        last = cin
        for j, c in enumerate(mlp):
            conv("%s.mlp_convs.%d" % (name, j), last, c, 1)
            bn("%s.mlp_bns.%d" % (name, j), c)
            last = c
    conv("conv1", 128, 128, 1)
    bn("bn1", 128)
    conv("conv2", 128, NUM_CLASSES, 1)
    return sd


def gcn_state_dict(seed, n_blocks, block="res", conv="edge"):
    """Seeded random weights for DenseDeepGCN(opt) (n_filters=64, k=16, in_channels=9, 13 classes) with the reference's
    state_dict keys and shapes for the given `block` / `conv` switches (ResGCN/sem_seg_dense/architecture.py:20-45):
    He-scaled conv weights, small biases and non-trivial eval BatchNorm statistics."""
    rng = np.random.RandomState(int(seed))
    sd = {}

    def basic(name, cin, cout, norm=True):
        sd[name + ".0.weight"] = (rng.standard_normal((cout, cin, 1, 1)) * np.sqrt(2.0 / cin)).astype(np.float32)
        sd[name + ".0.bias"] = rng.uniform(-0.1, 0.1, cout).astype(np.float32)
        if norm:   # BasicConv = Conv2d, ReLU, BatchNorm2d  -> the norm is sub-module 2
            sd[name + ".2.weight"] = rng.uniform(0.8, 1.2, cout).astype(np.float32)
            sd[name + ".2.bias"] = rng.uniform(-0.1, 0.1, cout).astype(np.float32)
            sd[name + ".2.running_mean"] = rng.uniform(-0.1, 0.1, cout).astype(np.float32)
            sd[name + ".2.running_var"] = rng.uniform(0.5, 1.5, cout).astype(np.float32)
            sd[name + ".2.num_batches_tracked"] = np.array(1, np.int64)

    basic("head.gconv.nn", 18, 64)
    for i in range(n_blocks - 1):
        cin = 64 * (i + 1) if block == "dense" else 64
        basic("backbone.%d.body.gconv.nn" % i, 2 * cin, 64)
    fdim = 64 * n_blocks * (n_blocks + 1) // 2 if block == "dense" else 64 * n_blocks
    basic("fusion_block", fdim, 1024)
    basic("prediction.0", fdim + 1024, 512)
    basic("prediction.1", 512, 256)
    basic("prediction.3", 256, NUM_CLASSES, norm=False)
    return sd


def gcn28_state_dict():
    """State dict of the 28-block ResGCN used by the BASELINE-size fixture (tests/golden/gcn28_room.npz) and by bench.py's
    resgcn workload: gcn_state_dict(7, 28) for the two big matrices (fusion_block.0.weight, prediction.0.0.weight: 13 MB
    that stay a seeded recipe) and, for everything else, the values a short fit of the reference network produced
    (tests/golden/gcn28_weights_small.npz, written by tests/golden/make_golden_big.py: fit_gcn28).  Random residual blocks
    alone blow the features up by 1e5 over 28 blocks."""
    import os
    sd = gcn_state_dict(7, 28)
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "gcn28_weights_small.npz")
    small = np.load(path)
    for k in small.files:
        assert k in sd and sd[k].shape == small[k].shape, k
        sd[k] = small[k]
    return sd


def gcn28_fit_state_dict():
    """The 28-block ResGCN weights of the free-running outcome fixture (tests/golden/gcn28_nb_outcome_fit*.npz, round 6):
    gcn_state_dict(7, 28) for the two big matrices and tests/golden/gcn28_weights_fit.npz for everything else - the reference
    network fitted IN EVAL MODE with small residual branches until it is right on ~0.9 of the rule labels of a 4096-point
    room (tests/golden/make_golden_big.py: fit_gcn28_frozen).  gcn28_state_dict() above (70 steps, accuracy 0.1 - 0.2, the
    reference chaotic against itself) stays what the teacher-forced 28-block fixture and bench.py use."""
    import os
    sd = gcn_state_dict(7, 28)
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "gcn28_weights_fit.npz")
    small = np.load(path)
    for k in small.files:
        assert k in sd and sd[k].shape == small[k].shape, k
        sd[k] = small[k]
    return sd


# RandLA-Net for S3DIS (RandLA-Net/helper_tool.py:41-60 ConfigS3DIS, RandLANet.py:150-190 of the reference)
RANDLA_D_OUT = (16, 64, 128, 256, 512)
RANDLA_RATIOS = (4, 4, 4, 4, 2)


def randla_layer_specs(d_out=RANDLA_D_OUT, n_classes=NUM_CLASSES):
    """[(name, cin, cout, has_bn)] of every weight layer in forward order.  Names follow the reference's scopes
    (RandLANet.py:150-190, 323-345, 396-410): fc0; Encoder_layer_i{mlp1, LFAmlp1, LFAatt_pooling_1fc, LFAatt_pooling_1mlp,
    LFAmlp2, LFAatt_pooling_2fc, LFAatt_pooling_2mlp, mlp2, shortcut}; decoder_0; Decoder_layer_j; fc1, fc2, fc."""
    specs = [("fc0", 6, 8, True)]
    d_in = 8
    for i, d in enumerate(d_out):
        p = "Encoder_layer_%d" % i
        specs += [(p + "mlp1", d_in, d // 2, True), (p + "LFAmlp1", 10, d // 2, True),
                  (p + "LFAatt_pooling_1fc", d, d, False), (p + "LFAatt_pooling_1mlp", d, d // 2, True),
                  (p + "LFAmlp2", d // 2, d // 2, True), (p + "LFAatt_pooling_2fc", d, d, False),
                  (p + "LFAatt_pooling_2mlp", d, d, True), (p + "mlp2", d, 2 * d, True), (p + "shortcut", d_in, 2 * d, True)]
        d_in = 2 * d
    specs.append(("decoder_0", d_in, d_in, True))
    enc_c = [2 * d_out[0]] + [2 * d for d in d_out]          # channels of f_encoder_list: enc_0, samp_0 .. samp_4
    feat = d_in
    for j in range(len(d_out)):
        skip = enc_c[-j - 2]
        specs.append(("Decoder_layer_%d" % j, skip + feat, skip, True))
        feat = skip
    specs += [("fc1", feat, 64, True), ("fc2", 64, 32, True), ("fc", 32, n_classes, False)]
    return specs


def randla_params(seed):
    """Seeded random parameters {name.weight [cout, cin], name.bias (absent for the attention fc), name.bn.{gamma,
    beta, mean, var}} for the network above (no checkpoint ships with the reference)."""
    rng = np.random.RandomState(int(seed))
    out = {}
    for name, cin, cout, bn in randla_layer_specs():
        out[name + ".weight"] = (rng.standard_normal((cout, cin)) * np.sqrt(1.0 / cin)).astype(np.float32)
        if "att_pooling" in name and name.endswith("fc"):
            continue                                          # tf.layers.dense(..., use_bias=False) (RandLANet.py:402)
        out[name + ".bias"] = rng.uniform(-0.1, 0.1, cout).astype(np.float32)
        if bn:
            out[name + ".bn.gamma"] = rng.uniform(0.8, 1.2, cout).astype(np.float32)
            out[name + ".bn.beta"] = rng.uniform(-0.1, 0.1, cout).astype(np.float32)
            out[name + ".bn.mean"] = rng.uniform(-0.1, 0.1, cout).astype(np.float32)
            out[name + ".bn.var"] = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    return out
