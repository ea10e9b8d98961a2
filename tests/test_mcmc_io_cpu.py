"""CPU tests of the reference's on-disk formats, the autocorrelation estimate and the stretch-move oracle."""
import json

import numpy as np
import pytest

from magprop_amd import mcmc_io
from oracle import stretch_oracle as so


def test_chain_files_follow_the_reference_format(tmp_path):
    """synth_mcmc.py:188-213: header 'Npars, Nwalk, Nstep', '%.6f, ' fields, lnprob last; per-parameter and lnp files."""
    rng = np.random.default_rng(0)
    chain = rng.normal(size=(3, 4, 6))
    lnp = rng.normal(size=(3, 4))
    base = str(tmp_path / "Humped")
    mcmc_io.write_chain_files(base, chain, lnp)
    lines = open(base + "_chain.csv").read().splitlines()
    assert lines[0] == "6, 4, 3" and len(lines) == 1 + 3 * 4
    want = "".join(f"{chain[0, 0, k]:.6f}, " for k in range(6)) + f"{lnp[0, 0]:.6f}"
    assert lines[1] == want
    assert lines[1 + 4] == "".join(f"{chain[1, 0, k]:.6f}, " for k in range(6)) + f"{lnp[1, 0]:.6f}"   # step-major
    p2 = open(base + "_2.csv").read().splitlines()
    assert len(p2) == 3 and p2[1] == ", ".join(f"{chain[1, i, 2]:.6f}" for i in range(4))
    lp = open(base + "_lnp.csv").read().splitlines()
    assert lp[2] == ", ".join(f"{lnp[2, i]:.6f}" for i in range(4))
    c2, l2 = mcmc_io.read_chain_file(base + "_chain.csv")
    assert np.allclose(c2, chain, atol=5e-7) and np.allclose(l2, lnp, atol=5e-7)


def test_dataset_and_info_files(tmp_path, gsynth):
    x, y, yerr = gsynth["Humped_x"], gsynth["Humped_y"], gsynth["Humped_yerr"]
    p = str(tmp_path / "Humped.csv")
    mcmc_io.write_dataset(p, x, y, yerr)
    assert open(p).readline().strip() == "x,y,yerr"                      # generate_data.py:70-71
    x2, y2, e2 = mcmc_io.read_dataset(p)
    assert np.array_equal(x2, x) and np.array_equal(y2, y) and np.array_equal(e2, yerr)
    q = tmp_path / "grb.csv"
    q.write_text("t,Lum50,Lum50err\n1.5,0.2,0.05\n3.0,0.1,0.02\n")       # k-corrected real-GRB columns
    t, l, e = mcmc_io.read_dataset(str(q))
    assert t.tolist() == [1.5, 3.0] and l.tolist() == [0.2, 0.1] and e.tolist() == [0.05, 0.02]
    (tmp_path / "bad.csv").write_text("a,b\n1,2\n")
    with pytest.raises(ValueError):
        mcmc_io.read_dataset(str(tmp_path / "bad.csv"))
    info = mcmc_io.write_info(str(tmp_path / "i.json"), 6, 24, 50, 1234, np.array([0.3, 0.5]), np.array([10.0, 12.0]))
    assert json.load(open(tmp_path / "i.json")) == info == {"Npars": 6, "Nwalk": 24, "Nstep": 50, "seed": 1234,
                                                            "acceptance_fraction": 0.4, "tau": [10.0, 12.0]}


def test_integrated_time_of_ar1():
    """AR(1) with coefficient rho has tau = (1+rho)/(1-rho)."""
    rng = np.random.default_rng(1)
    rho = 0.8
    n, w = 20000, 16
    x = np.zeros((n, w, 1))
    e = rng.normal(size=(n, w))
    for i in range(1, n):
        x[i, :, 0] = rho * x[i - 1, :, 0] + e[i]
    tau = mcmc_io.integrated_time(x)
    assert abs(tau[0] - (1 + rho) / (1 - rho)) < 0.8
    with pytest.raises(RuntimeError):
        mcmc_io.integrated_time(x[:100], quiet=False)


def test_philox_known_answer():
    """Philox4x32-10 known-answer vectors of the Random123 distribution (kat_vectors)."""
    assert so.philox4x32_10(0, 0, 0, 0, 0, 0) == (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)
    assert so.philox4x32_10(0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff) == \
        (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)
    assert so.philox4x32_10(0xa4093822, 0x299f31d0, 0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344) == \
        (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)


def test_stretch_move_samples_the_target():
    """The move as restated in the oracle leaves a unit Gaussian invariant (mean, variance, acceptance)."""
    rng = np.random.default_rng(2)
    pos = rng.normal(size=(40, 3)) * 3.0 + 2.0          # start far from equilibrium
    chain, lnp, acc = so.run(pos, 600, seed=99)
    tail = chain[200:].reshape(-1, 3)
    assert np.all(np.abs(tail.mean(axis=0)) < 0.15)
    assert np.all(np.abs(tail.var(axis=0) - 1.0) < 0.2)
    assert 0.3 < acc.mean() / 600 < 0.8
    p = so.split(99, 5, 0, 40)
    assert sorted(p) == list(range(40)) and p != list(range(40))
