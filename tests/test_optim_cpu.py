"""bitorch_engine.optim (DiodeMix, GaLoreProjector) against the reference's own optimiser run on CPU tensors: tests/golden/optim_diodemix.npz,
written by oracle/gen_golden.py (optim_diodemix_vectors: the imported reference, optim/diode_beta.py:37-196 + optim/galore_projector.py:17-124).
Host logic on both sides (the reference's optimiser is pure torch): CPU tests.  The MPQ parameter's step needs the HIP unpack / pack kernels:
tests/test_gpu_parity.py::test_diodemix_steps_an_mpq_parameter_like_the_reference."""
import os

import numpy as np
import pytest
import torch

from bitorch_engine.optim import DiodeMix, GaLoreProjector


@pytest.fixture(scope="module")
def vec(golden_dir):
    return np.load(os.path.join(golden_dir, "optim_diodemix.npz"))


def test_float_parameters_take_the_reference_adamw_steps(vec):
    """Two groups with their own lr / weight decay / bias correction, four steps: every parameter after every step and the moments at the end.
    Elementwise fp32 arithmetic in the reference's op order: equal to the last bit on the machine that wrote the vectors, 1e-6 relative anywhere
    (a CPU kernel may or may not contract a multiply-add)."""
    ps = {n: torch.nn.Parameter(torch.from_numpy(vec[f"float_{n}_0"]).clone()) for n in ("p1", "p2", "p3")}
    opt = DiodeMix([{"params": [ps["p1"], ps["p2"]], "weight_decay": 0.01}, {"params": [ps["p3"]], "lr": 5e-4, "correct_bias": False}], lr=1e-3, betas=(0.9, 0.99), eps=1e-6)
    for it in range(1, 5):
        for n, p in ps.items():
            p.grad = torch.from_numpy(vec[f"float_{n}_grad{it}"]).clone()
        opt.step()
        for n, p in ps.items():
            torch.testing.assert_close(p.detach(), torch.from_numpy(vec[f"float_{n}_{it}"]), rtol=1e-6, atol=1e-7, msg=lambda m: f"{n} after step {it}: {m}")
    st = opt.state[ps["p1"]]
    assert set(st) == {"step", "exp_avg_l", "exp_avg_s"} and float(st["step"]) == 4.0
    torch.testing.assert_close(st["exp_avg_l"], torch.from_numpy(vec["float_p1_m"]), rtol=1e-6, atol=1e-9)
    torch.testing.assert_close(st["exp_avg_s"], torch.from_numpy(vec["float_p1_v"]), rtol=1e-6, atol=1e-12)
    assert not torch.equal(ps["p3"].detach(), torch.from_numpy(vec["float_p3_0"]))


@pytest.mark.parametrize("tag,pt,shape", [("std_tall", "std", (48, 20)), ("std_wide", "std", (20, 48)), ("rstd_tall", "reverse_std", (48, 20)),
                                           ("rstd_wide", "reverse_std", (20, 48)), ("left", "left", (20, 48)), ("right", "right", (20, 48)), ("full", "full", (20, 48))])
def test_galore_groups_follow_the_reference(vec, tag, pt, shape):
    """A GaLore group (rank 4, projector refreshed every second step, scale 0.25) of every projection type: the parameter after each of four
    steps.  The SVD is LAPACK's on both sides; 1e-4 relative leaves room for another CPU's BLAS, a wrong side or a missing scale is off by 1."""
    p = torch.nn.Parameter(torch.from_numpy(vec[f"galore_{tag}_0"]).clone())
    assert tuple(p.shape) == shape
    opt = DiodeMix([{"params": [p], "rank": 4, "update_proj_gap": 2, "scale": 0.25, "proj_type": pt}], lr=2e-3, betas=(0.9, 0.99), weight_decay=0.0)
    for it in range(1, 5):
        p.grad = torch.from_numpy(vec[f"galore_{tag}_grad{it}"]).clone()
        opt.step()
        torch.testing.assert_close(p.detach(), torch.from_numpy(vec[f"galore_{tag}_{it}"]), rtol=1e-4, atol=1e-6, msg=lambda m: f"{tag} after step {it}: {m}")
    pr = opt.state[p]["projector"]
    assert isinstance(pr, GaLoreProjector) and pr.rank == 4 and pr.proj_type == pt
    m = opt.state[p]["exp_avg_l"]
    assert 4 in m.shape and m.numel() < p.numel(), "the moments live in the projected space"


def test_galore_projector_sides_and_errors():
    g = torch.Generator().manual_seed(5)
    G = torch.randn((12, 30), generator=g)
    for pt, low_shape in (("std", (3, 30)), ("reverse_std", (12, 3)), ("left", (3, 30)), ("right", (12, 3)), ("full", (3, 3))):
        pr = GaLoreProjector(3, update_proj_gap=10, scale=2.0, proj_type=pt)
        low = pr.project(G, 0)
        assert tuple(low.shape) == low_shape, pt
        first = pr.ortho_matrix
        pr.project(G * 2, 7)  # not a refresh step: the basis is kept
        assert pr.ortho_matrix is first
        pr.project(G * 2, 10)
        assert pr.ortho_matrix is not first
        back = GaLoreProjector(3, scale=2.0, proj_type=pt)
        back.project(G, 0)
        full = back.project_back(low)
        assert tuple(full.shape) == (12, 30)
        # projecting the reconstruction again gives scale x the low-rank gradient (orthonormal factors)
        torch.testing.assert_close(back.project(full, 1), 2.0 * low, rtol=1e-4, atol=1e-5)
    half = GaLoreProjector(2, proj_type="right").get_orthogonal_matrix(G.half(), 2, "right")
    assert half.dtype == torch.half and tuple(half.shape) == (2, 30)
    with pytest.raises(ValueError, match="left, right or full"):
        GaLoreProjector(2).get_orthogonal_matrix(G, 2, "up")


@pytest.mark.parametrize("tag,dtype", [("f32", torch.float), ("bf16", torch.bfloat16)])
@pytest.mark.parametrize("kind", ["binlin", "nbitlin"])
def test_quantised_parameters_step_like_the_reference(vec, tag, dtype, kind):
    """A binary linear parameter (sign flips) and a W4A4-style integer parameter through the optimiser, three steps, moments in fp32 and bf16.
    The sign carrier's first short average is drawn with torch.rand_like at the first step: under the same torch.manual_seed the mirror consumes
    the generator exactly as the reference does, so data, data dtype and both moments agree bit for bit."""
    from bitorch_engine.layers.qlinear.binary import BinaryLinearParameter
    from bitorch_engine.layers.qlinear.nbit import nBitLinearParameter
    cls = BinaryLinearParameter if kind == "binlin" else nBitLinearParameter
    key = f"q_{tag}_{kind}"
    cast = (lambda a: torch.from_numpy(a)) if dtype == torch.float else (lambda a: torch.from_numpy(a.view(np.int16).copy()).view(dtype))
    w0 = torch.from_numpy(vec[key + "_w0"])
    p = cls(w0.clone(), requires_grad=False)
    opt = DiodeMix([p], lr=3e-2, betas=(0.9, 0.99), eps=1e-6, weight_decay=0.0, dtype=dtype)
    torch.manual_seed(4242)
    for it in range(1, 4):
        p.grad = None
        p.grad_dtype = None
        p.grad = torch.from_numpy(vec[f"{key}_grad{it}"]).clone()
        opt.step()
        assert str(p.data.dtype) == str(vec[f"{key}_wdtype{it}"][0]), f"step {it}: data dtype"
        want = vec[f"{key}_w{it}"]
        got = p.data
        if got.dtype in (torch.float16, torch.bfloat16):
            assert torch.equal(got, torch.from_numpy(want.view(np.int16).copy()).view(got.dtype)), f"step {it}: data"
        else:
            assert np.array_equal(got.numpy(), want), f"step {it}: data"
        st = opt.state[p]
        assert st["exp_avg_s"].dtype == dtype
        assert torch.equal(st["exp_avg_s"], cast(vec[f"{key}_exp_s{it}"])), f"step {it}: short / second moment"
        assert torch.equal(st["exp_avg_l"], cast(vec[f"{key}_exp_l{it}"])), f"step {it}: long / first moment"
    assert float(opt.state[p]["step"]) == 3.0
    assert not np.array_equal(p.data.float().numpy(), w0.float().numpy()), "three steps changed nothing: vacuous vectors"


def test_constructor_checks_and_skipped_parameters():
    w = torch.nn.Parameter(torch.ones(3))
    for kw, msg in (({"lr": -1.0}, "learning rate"), ({"betas": (1.0, 0.5)}, "beta parameter"), ({"betas": (0.5, -0.1)}, "beta parameter"), ({"eps": -1e-3}, "epsilon")):
        with pytest.raises(ValueError, match=msg):
            DiodeMix([w], **kw)
    opt = DiodeMix([w])
    assert opt.defaults == {"lr": 1e-4, "betas": (0.99, 0.9999), "eps": 1e-6, "weight_decay": 0.0, "correct_bias": True} and opt.dtype == torch.float
    opt.step()  # no gradient: nothing happens, no state appears
    assert len(opt.state) == 0 and torch.equal(w.detach(), torch.ones(3))
    assert opt.step(lambda: torch.tensor(3.5)) == 3.5
    w.grad = torch.ones(3).to_sparse()
    with pytest.raises(RuntimeError, match="sparse"):
        opt.step()


def test_state_dict_round_trip_continues_the_same_trajectory(vec):
    """Optimiser checkpoints: state_dict() after two steps, loaded into a fresh optimiser over copies of the parameters, then two more steps ==
    four uninterrupted steps -- float group and GaLore group (the projector object travels inside the state, as in the reference)."""
    def make():
        a = torch.nn.Parameter(torch.from_numpy(vec["float_p1_0"]).clone())
        b = torch.nn.Parameter(torch.from_numpy(vec["galore_std_wide_0"]).clone())
        opt = DiodeMix([{"params": [a], "weight_decay": 0.01}, {"params": [b], "rank": 4, "update_proj_gap": 2, "scale": 0.25, "proj_type": "std"}], lr=1e-3, betas=(0.9, 0.99))
        return a, b, opt

    def feed(a, b, it):
        a.grad = torch.from_numpy(vec[f"float_p1_grad{it}"]).clone()
        b.grad = torch.from_numpy(vec[f"galore_std_wide_grad{it}"]).clone()

    a, b, opt = make()
    for it in (1, 2, 3, 4):
        feed(a, b, it)
        opt.step()
    a2, b2, opt2 = make()
    for it in (1, 2):
        feed(a2, b2, it)
        opt2.step()
    sd = opt2.state_dict()
    a3, b3, opt3 = make()
    with torch.no_grad():
        a3.copy_(a2)
        b3.copy_(b2)
    opt3.load_state_dict(sd)
    assert isinstance(opt3.state[b3]["projector"], GaLoreProjector) and float(opt3.state[a3]["step"]) == 2.0
    for it in (3, 4):
        feed(a3, b3, it)
        opt3.step()
    assert torch.equal(a3.detach(), a.detach()) and torch.equal(b3.detach(), b.detach())
