"""Deterministic DCP weights shared by tests/golden/make_golden_dcp.py (which loads them into the REFERENCE model) and
by the tests (which load them into houv_amd.models.dcp.Model): the repository ships no trained DCP checkpoint, and
regenerating 22 MB of weights from a seed keeps the golden file small.  Names/shapes = the reference's state_dict
(registration/models/dcp.py:269-391)."""
import numpy as np


def _attn(prefix):
    out = []
    for i in range(4):
        out += [(f"{prefix}.linears.{i}.weight", (512, 512)), (f"{prefix}.linears.{i}.bias", (512,))]
    return out


def _ff(prefix):
    return [(f"{prefix}.w_1.weight", (1024, 512)), (f"{prefix}.w_1.bias", (1024,)),
            (f"{prefix}.w_2.weight", (512, 1024)), (f"{prefix}.w_2.bias", (512,))]


def _ln(prefix):
    return [(f"{prefix}.a_2", (512,)), (f"{prefix}.b_2", (512,))]


def spec():
    s = [("emb_nn.conv1.weight", (64, 6, 1, 1)), ("emb_nn.conv2.weight", (64, 64, 1, 1)),
         ("emb_nn.conv3.weight", (128, 64, 1, 1)), ("emb_nn.conv4.weight", (256, 128, 1, 1)),
         ("emb_nn.conv5.weight", (512, 512, 1, 1))]
    for i, c in zip(range(1, 6), (64, 64, 128, 256, 512)):
        s += [(f"emb_nn.bn{i}.weight", (c,)), (f"emb_nn.bn{i}.bias", (c,)), (f"emb_nn.bn{i}.running_mean", (c,)),
              (f"emb_nn.bn{i}.running_var", (c,))]
    e, d = "pointer.model.encoder", "pointer.model.decoder"
    s += _attn(f"{e}.layers.0.self_attn") + _ff(f"{e}.layers.0.feed_forward")
    s += _ln(f"{e}.layers.0.sublayer.0.norm") + _ln(f"{e}.layers.0.sublayer.1.norm") + _ln(f"{e}.norm")
    s += _attn(f"{d}.layers.0.self_attn") + _attn(f"{d}.layers.0.src_attn") + _ff(f"{d}.layers.0.feed_forward")
    s += (_ln(f"{d}.layers.0.sublayer.0.norm") + _ln(f"{d}.layers.0.sublayer.1.norm")
          + _ln(f"{d}.layers.0.sublayer.2.norm") + _ln(f"{d}.norm"))
    return s


def make_state(seed=1234):
    rng = np.random.default_rng(seed)
    st = {}
    for name, shape in spec():
        if name.endswith("running_var") or name.endswith(".a_2") or (".bn" in name and name.endswith("weight")):
            v = rng.uniform(0.5, 1.5, shape)
        elif name.endswith("running_mean") or name.endswith(".b_2") or (".bn" in name and name.endswith("bias")):
            v = rng.normal(0, 0.2, shape)
        else:
            fan_in = shape[1] if len(shape) > 1 else 512
            v = rng.uniform(-1, 1, shape) / np.sqrt(fan_in)
        st[name] = v.astype(np.float32)
    return st


def make_pointnet_state(seed=99):
    """Seeded weights for dcp.py's PointNet embedding (:246-258)."""
    rng = np.random.default_rng(seed)
    dims = (3, 64, 64, 64, 128, 512)
    st = {}
    for i in range(5):
        st[f"conv{i + 1}.weight"] = (rng.uniform(-1, 1, (dims[i + 1], dims[i], 1)) / np.sqrt(dims[i])).astype(np.float32)
        st[f"bn{i + 1}.weight"] = rng.uniform(0.5, 1.5, dims[i + 1]).astype(np.float32)
        st[f"bn{i + 1}.bias"] = rng.normal(0, 0.2, dims[i + 1]).astype(np.float32)
        st[f"bn{i + 1}.running_mean"] = rng.normal(0, 0.2, dims[i + 1]).astype(np.float32)
        st[f"bn{i + 1}.running_var"] = rng.uniform(0.5, 1.5, dims[i + 1]).astype(np.float32)
    return st
