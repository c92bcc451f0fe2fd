"""End to end: the EM driver's loop (est_params_histories.cpp:232-272: reset, run_mcmc, M-step)
on the GPU versus the same loop over the oracle in REFERENCE-schedule mode (rung A, which is
the reference bit for bit).  Different random streams, so the comparison is statistical: the
parameter estimates of both trajectories agree within Monte-Carlo error, from the true
parameters and from a perturbed start."""
import numpy as np
import pytest

import orc
from common import simulate, _tmp
from epievo_amd import host

pytestmark = pytest.mark.gpu

N_SITES, ITERS, BURN, BATCH = 30000, 5, 5, 10


def _em(engine_reset, engine_run, set_model, model, tree):
    traj = []
    for it in range(ITERS):
        set_model(model)
        engine_reset()
        J, D, acc = engine_run(it)
        model, _, llh, _ = host.m_step(model, tree.branches, J, D)
        traj.append((model.rates.copy(), acc))
    return traj


@pytest.mark.parametrize("start", ["truth", "perturbed"])
def test_em_trajectories_agree(start):
    from epievo_amd.sampler import DeviceSampler
    truth, tree, fp = simulate("tree", N_SITES, seed=21)
    m0 = truth if start == "truth" else host.Model.read(
        _tmp("perturbed.param", "stationary\t0.80\t0.86\nbaseline\t-0.5\t-1.5\n"), scale=True)
    d = DeviceSampler(0)
    d.set_tree(tree)
    d.set_model(m0)
    d.upload_paths(fp, 24)

    def gpu_run(it):
        J, D, nacc = d.run_mcmc(BURN, BATCH, 77, it * (BURN + BATCH))
        return J, D, nacc / float(BATCH * (N_SITES - 2))
    gpu = _em(d.reset, gpu_run, d.set_model, m0, tree)

    o = orc.Oracle(tree, m0, fp, "A", seed=5)

    def cpu_run(it):
        J, D, nacc, acc = o.run_mcmc(BURN, BATCH)
        return J, D, acc
    cpu = _em(o.reset, cpu_run, o.set_model, m0, tree)

    for (rg, ag), (rc, ac) in zip(gpu, cpu):
        assert np.allclose(rg, rc, rtol=0.06), (rg, rc)
        assert abs(ag - ac) < 0.01
    # and both end near the generating parameters
    assert np.allclose(gpu[-1][0], truth.rates, rtol=0.25)
