"""Host-side mirror of the reference's class protocol (CPU only: no kernel is launched here)."""
import ctypes
import os
import re
import warnings

import numpy as np
import pytest
import torch

import bayeformers_amd as bf
import bayeformers_amd.nn as bnn
from bayeformers_amd import _C


def _net():
    torch.manual_seed(7)
    return torch.nn.Sequential(torch.nn.Linear(12, 9), torch.nn.Tanh(), torch.nn.Linear(9, 4, bias=False))


@pytest.mark.parametrize("mode", ["moped", "plain"])
def test_to_bayesian_matches_reference_bitwise(golden_dir, mode):
    """Same keys, same requires_grad flags, bit-identical mu/rho/prior values as the reference's to_bayesian
    (fixtures from /root/reference/bayeformers/__init__.py:19-63 + layers/linear.py:106-165)."""
    g = np.load(f"{golden_dir}/conversion.npz")
    net = _net()
    torch.manual_seed(8)
    b = bf.to_bayesian(net, delta=0.1, freeze=True) if mode == "moped" else bf.to_bayesian(net)
    sd = b.state_dict()
    assert sorted(sd.keys()) == list(g[f"{mode}_keys"])
    assert sorted(n for n, p in b.named_parameters() if p.requires_grad) == list(g[f"{mode}_requires_grad"])
    for k, v in sd.items():
        assert np.array_equal(v.numpy(), g[f"{mode}/{k}"]), k


def test_moped_shares_storage_and_minus_inf_rule():
    lin = torch.nn.Linear(6, 3)
    with torch.no_grad():
        lin.weight[0, 0] = 0.0
    b = bnn.Linear.from_frequentist(lin, delta=0.05, freeze=True)
    assert b.weight.mu.data_ptr() == lin.weight.data_ptr() == b.weight_prior.mu.data_ptr()
    assert b.weight.rho[0, 0].item() == 0.0 and not b.weight.mu.requires_grad and b.weight.rho.requires_grad
    assert torch.all(b.weight_prior.rho == 1.0)
    assert isinstance(b.bias_prior, bnn.Gaussian)


def test_exact_class_lookup_only():
    class MyLinear(torch.nn.Linear):
        pass

    m = torch.nn.Sequential(MyLinear(3, 3), torch.nn.Linear(3, 2))
    b = bf.to_bayesian(m)
    assert isinstance(b.model[0], MyLinear) and isinstance(b.model[1], bnn.Linear)
    assert m[1].__class__ is torch.nn.Linear  # the source model is untouched (deepcopy)


def test_state_dict_round_trip_and_shared_default_prior():
    b = bf.to_bayesian(_net())
    assert b.model[0].weight_prior is bnn.DEFAULT_SCALED_GAUSSIAN_MIXTURE is b.model[2].weight_prior
    sd = {k: v.clone() for k, v in b.state_dict().items()}
    b2 = bf.to_bayesian(_net())
    b2.load_state_dict(sd)
    for k, v in b2.state_dict().items():
        assert torch.equal(v, sd[k])
    assert np.allclose(bnn.DEFAULT_SCALED_GAUSSIAN_MIXTURE.constants(), (0.5, 1.0, 0.0024787522852420807))


def test_dtype_casts_keep_fp32_masters():
    b = bf.to_bayesian(torch.nn.Sequential(torch.nn.Linear(4, 4), torch.nn.LayerNorm(4)), delta=0.05)
    b = b.to(torch.bfloat16)
    assert b.model[1].weight.dtype == torch.bfloat16
    for p in (b.model[0].weight.mu, b.model[0].weight.rho, b.model[0].weight_prior.rho, b.model[0].bias.mu):
        assert p.dtype == torch.float32


def test_forward_on_cpu_raises_instead_of_falling_back():
    b = bf.to_bayesian(_net())
    with pytest.raises(_C.BayeFormersAMDError, match="ROCm device"):
        b(torch.randn(2, 12))
    with pytest.raises(_C.BayeFormersAMDError):
        b.model[0].weight.sample()


def test_model_protocol():
    assert bnn.TORCH2BAYE == {torch.nn.Linear: bnn.Linear}
    assert bnn.is_module_bayesian(bnn.Linear(2, 2)) and not bnn.is_module_bayesian(torch.nn.Linear(2, 2))
    empty = bnn.Model(torch.nn.ReLU())
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert empty.log_prior() == 0.0 and empty.log_variational_posterior() == 0.0
        assert len(w) == 2
    with pytest.raises(NotImplementedError):
        bnn.Model().forward(1)
    with pytest.raises(NotImplementedError):
        bnn.Parameter().sample()
    with pytest.raises(NotImplementedError):
        bnn.Initialization()(None, None)
    assert bnn.NoneParameter().sample() is None and bnn.NoneParameter().log_prob(None) == 0.0
    b = bf.to_bayesian(_net())
    assert [l.layer_id for l in b.fused_children()] == [0, 1]
    assert len(list(b.bayesian_children)) == 2
    assert float(b.log_prior()) == 0.0  # nothing ran yet, as in the reference


def test_header_and_library_export_the_same_symbols():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "bayeformers_amd.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(bf_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_C.SYMBOLS), declared ^ set(_C.SYMBOLS)
    lib = ctypes.CDLL(_C.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert _C.lib().bf_version() == _C.ABI_VERSION == 6


def test_struct_layout_matches_header():
    # 4 + 3*4 + 5*8 = 56 bytes; tensor = 8+8+8+56+4+4+8 = 96 bytes (natural alignment, as the C compiler lays it out)
    assert ctypes.sizeof(_C.bf_prior_t) == 56 and ctypes.sizeof(_C.bf_tensor_t) == 96


def test_error_reporting_without_gpu():
    lib = _C.lib()
    rc = lib.bf_sample_logprob(None, 1, 1, 0, 0, None, None, 0, None)
    assert rc != 0 and b"tensors is NULL" in lib.bf_last_error()
    assert lib.bf_linear_fwd_workspace_bytes(10, 32, 768, 768, 1, _C.BF_DT_BF16, _C.BF_DT_F32) > 10 * 768 * 768 * 2


def test_header_is_plain_c_and_every_entry_links_from_c(tmp_path):
    """The boundary is a C ABI: include/bayeformers_amd.h compiles as C99 with gcc and a C translation unit that takes
    the address of every declared entry point links against the shared library (no compute call: no GPU here)."""
    import shutil
    import subprocess

    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("gcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = os.path.join(root, "include", "bayeformers_amd.h")
    names = sorted(set(re.findall(r"\b(bf_[a-z0-9_]+)\s*\(", open(header).read())) & set(_C.SYMBOLS))
    assert len(names) == len(_C.SYMBOLS)
    src = tmp_path / "link_check.c"
    body = "\n".join(f"    p[{i}] = (fn)&{n};" for i, n in enumerate(names))
    src.write_text('#include "bayeformers_amd.h"\n#include <stdio.h>\ntypedef void (*fn)(void);\nint main(void) {\n'
                   f"    fn p[{len(names)}];\n    int i, n = 0;\n{body}\n"
                   f"    for (i = 0; i < {len(names)}; ++i) n += p[i] != 0;\n"
                   '    printf("%d %d\\n", bf_version(), n);\n    return 0;\n}\n')
    exe = tmp_path / "link_check"
    libdir = os.path.dirname(_C.LIB_PATH)
    subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(root, "include"), str(src),
                    "-L", libdir, "-lbayeformers_amd", f"-Wl,-rpath,{libdir}", "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    assert out[-1] == str(len(names)) and int(out[0]) == _C.lib().bf_version()
    # the developer build's header (csrc/bf_dev_api.h) is C too, finds the public header by itself, and declares
    # exactly the developer symbols the binding knows
    dev = os.path.join(root, "bayeformers_amd", "csrc", "bf_dev_api.h")
    tu = tmp_path / "dev_check.c"
    tu.write_text(f'#include "{dev}"\nint main(void) {{ return 0; }}\n')
    subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-pedantic", "-fsyntax-only", str(tu)], check=True)
    assert set(re.findall(r"\b(bf_[a-z0-9_]+)\s*\(", open(dev).read())) == set(_C.DEV_SYMBOLS)


def test_recompute_context_is_bound_to_the_graph_that_runs_backward():
    """A checkpointed block recomputed during backward must replay the sample indices of the forward that backward belongs
    to — not those of whatever bnn.Model forward finished last (an evaluation pass, another model, a second loss)."""
    import torch.utils.checkpoint as cp

    from bayeformers_amd import random as bfr
    from bayeformers_amd.nn.model import Model

    seen = []

    class Probe(torch.nn.Module):
        """Stands in for a Bayesian layer: outside a running forward it asks for the recompute context, like
        bnn.Linear.forward does, and records the sample base it would draw epsilon from."""

        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.ones(3))

        def forward(self, x):
            ctx = bfr.STATE.ctx if bfr.STATE.ctx is not None else bfr.recompute_context()
            seen.append(None if ctx is None else ctx.sample_base)
            return x * self.w

    class Net(Model):
        def __init__(self):
            super().__init__()
            self.p = Probe()

        def forward(self, x):
            return cp.checkpoint(self.p, x, use_reentrant=False)

    bfr.manual_seed(5)
    a, b = Net(), Net()
    x = torch.randn(2, 3, requires_grad=True)
    ya = a(x)            # sample base 0
    with torch.no_grad():
        a(x)             # an evaluation pass in between: base 1, must not be picked up
    yb = b(x)            # another model: base 2
    del seen[:]
    ya.sum().backward()  # recomputes a's block
    assert seen == [0], seen
    del seen[:]
    yb.sum().backward()
    assert seen == [2], seen
    # two forwards of checkpointed models in ONE backward: ambiguous -> loud
    y1, y2 = a(x), b(x)
    with pytest.raises(RuntimeError, match="could own it"):
        (y1.sum() + y2.sum()).backward()
    # a finished forward keeps no activations
    assert all(c.shared_out == {} for c in bfr.STATE.live_ctxs) and len(bfr.STATE.live_ctxs) <= bfr.LIVE_CONTEXTS


def test_recompute_context_with_outputs_the_walk_does_not_know():
    """a checkpointed model whose output is a dataclass (walked) or an opaque object (not walkable: no stamped
    hook) must still find its forward in EVERY training step, not only while one finished forward is remembered."""
    import dataclasses

    import torch.utils.checkpoint as cp

    from bayeformers_amd import random as bfr
    from bayeformers_amd.nn.model import Model

    seen = []

    class Probe(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.ones(3))

        def forward(self, x):
            ctx = bfr.STATE.ctx if bfr.STATE.ctx is not None else bfr.recompute_context()
            seen.append(None if ctx is None else ctx.sample_base)
            return x * self.w

    @dataclasses.dataclass
    class Out:
        logits: torch.Tensor
        extra: dict

    class Opaque:
        __slots__ = ("t",)

        def __init__(self, t):
            self.t = t

    class Net(Model):
        def __init__(self, wrap):
            super().__init__()
            self.p, self.wrap = Probe(), wrap

        def forward(self, x):
            return self.wrap(cp.checkpoint(self.p, x, use_reentrant=False))

    x = torch.randn(2, 3, requires_grad=True)
    for wrap, get in ((lambda t: Out(t, {"k": [t]}), lambda o: o.logits), (Opaque, lambda o: o.t)):
        bfr.manual_seed(11)
        net = Net(wrap)
        for step in range(4):   # more steps than one: finished forwards pile up in live_ctxs
            out = net(x)
            del seen[:]
            get(out).sum().backward()
            assert seen == [step], (wrap, step, seen)


def test_moped_prior_alias_is_decided_on_contents_and_tracks_edits():
    """ops.prior_alias: the sampling kernel may skip the prior's mu / rho (8 instead of 16 bytes per scalar) exactly when
    the prior is N(the posterior's frozen mean, one constant sigma) — what to_bayesian(delta, freeze=True) builds
    (/root/reference/bayeformers/nn/layers/linear.py:140-150)."""
    import copy

    import bayeformers_amd as bf
    from bayeformers_amd import ops

    torch.manual_seed(3)
    net = torch.nn.Sequential(torch.nn.Linear(12, 9), torch.nn.Tanh(), torch.nn.Linear(9, 4))
    frozen = bf.to_bayesian(copy.deepcopy(net), delta=0.05, freeze=True)
    lin = frozen.model[0]
    sp = float(torch.nn.functional.softplus(torch.tensor(1.0)))
    assert ops.prior_alias(lin.weight, lin.weight_prior) == pytest.approx(sp, rel=1e-7)
    assert ops.prior_alias(lin.bias, lin.bias_prior) == pytest.approx(sp, rel=1e-7)
    # equal copies instead of one storage (what a device move leaves behind): still an alias
    lin.weight_prior.mu.data = lin.weight_prior.mu.data.clone()
    assert lin.weight_prior.mu.data_ptr() != lin.weight.mu.data_ptr()
    assert ops.prior_alias(lin.weight, lin.weight_prior) is not None
    # a trainable mean is never aliased
    trainable = bf.to_bayesian(copy.deepcopy(net), delta=0.05, freeze=False)
    assert ops.prior_alias(trainable.model[0].weight, trainable.model[0].weight_prior) is None
    # in-place edits (optimizer steps, load_state_dict) bump the version counter: seen, re-checked
    with torch.no_grad():
        lin.weight_prior.rho[0, 0] = 2.0
    assert ops.prior_alias(lin.weight, lin.weight_prior) is None
    with torch.no_grad():
        lin.weight_prior.rho[0, 0] = 1.0
        lin.weight.mu[1, 1] += 1.0
    assert ops.prior_alias(lin.weight, lin.weight_prior) is None
    with torch.no_grad():
        lin.weight_prior.mu.copy_(lin.weight.mu)
    assert ops.prior_alias(lin.weight, lin.weight_prior) is not None


def test_bayesian_children_walk_is_kept_and_notices_a_swapped_layer():
    """Model.log_prior() asks for the Bayesian children on every call (model.py:70-78); the walk is kept between calls and
    thrown away when a kept child is no longer registered where it was found."""
    import bayeformers_amd as bf

    net = torch.nn.Sequential(torch.nn.Linear(8, 8), torch.nn.ReLU(), torch.nn.Linear(8, 4))
    b = bf.to_bayesian(net)
    first, only = b._children()
    assert len(first) == 2 and only and b._children()[0] is first          # kept
    assert list(b.bayesian_children) == first                              # the reference's property still walks
    b.model[2] = torch.nn.Linear(8, 4)                                     # a frequentist layer takes a Bayesian one's place
    again, only = b._children()
    assert len(again) == 1 and again is not first and only
    b.model[2] = bnn.Linear(8, 4)
    b.refresh()
    assert len(b._children()[0]) == 2


def test_fuse_ffn_pairs_rewires_only_transformer_feed_forward_pairs():
    import bayeformers_amd as bf
    from transformers import BertConfig, BertForSequenceClassification

    cfg = BertConfig(hidden_size=64, num_hidden_layers=2, num_attention_heads=2, intermediate_size=128, vocab_size=100,
                     max_position_embeddings=32)
    b = bf.to_bayesian(BertForSequenceClassification(cfg).eval(), delta=0.05)
    assert bf.fuse_activations(b) == 2
    assert bf.fuse_ffn_pairs(b) == 2 and bf.fuse_ffn_pairs(b) == 0          # idempotent
    layer = b.model.bert.encoder.layer[0]
    assert layer.feed_forward_chunk.__func__ is bf._ffn_pair_chunk and hasattr(layer, "_bf_plain_ffn_chunk")
    assert bf.fuse_ffn_pairs(bf.to_bayesian(torch.nn.Sequential(torch.nn.Linear(4, 4)))) == 0
