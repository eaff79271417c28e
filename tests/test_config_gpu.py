"""Every per-context option of the library (include/hgmm.h: hgmm_config_*; csrc/hgmm_ctx.h: ConfigKey) against the golden
of the path it selects.  An option is a product path: each one listed by hgmm_config_name() must have a case here (the test
fails when one is added without), and the cases are the reference-pinned tests of the default path run once more with
the option set -- same fixtures, same tolerances, so "the option's path == the reference" is held exactly as the default's is.

The options replace round 5's 46 environment switches read at call time: the library reads the environment once, in
hgmm_create (HGMM_<NAME> gives an option its start value), tuning knobs are fixed at their measured values and rejected
variants are gone with their code."""
import numpy as np
import pytest

import test_flat_gpu
import test_fullcov_gpu
import test_kmeans_gpu
import test_tree_batch_gpu
import test_tree_gpu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import hgmm_amd
    c = hgmm_amd.Context(0)
    yield c
    c.close()


def _flat(ctx, bunny):
    # e_step / predict of the reference's small golden (its own float32 output) + the bunny J = 800 frame of BASELINE configs[1]
    test_flat_gpu.test_estep_small_golden(ctx, "W", "diag")
    test_flat_gpu.test_estep_small_golden(ctx, "G", "diag")
    test_flat_gpu.test_estep_bunny(ctx, bunny, 800, "W", "diag")


def _tree(ctx, bunny):
    test_tree_gpu.test_build_matches_reference_golden(ctx, "hgmm_build_L2.npz")
    test_tree_gpu.test_build_matches_reference_golden(ctx, "hgmm_build_L3.npz")
    test_tree_gpu.test_build_bunny_subsample_L4_vs_oracle(ctx, bunny)


def _fullcov(ctx, bunny):
    test_fullcov_gpu.test_fullcov_matches_reference_golden(ctx, 8)
    test_fullcov_gpu.test_fullcov_matches_reference_golden(ctx, 32)


def _kmeans(ctx, bunny):
    for name in ("uniform", "blobs", "bunny"):
        test_kmeans_gpu.test_fit_matches_reference_init(ctx, bunny, name)


def _reg(ctx, bunny):
    # the registration loop with the 6 x 6 solve, the twist and the stop rule on the device: the reference's recorded
    # per-iteration transforms (1e-8), the Python path's trajectory and stopping iteration, budget split == one call;
    # batches of pairs bit for bit the serial call under the same option, an ill-conditioned pair leaving its batch
    test_tree_gpu.test_registration_loop_inside_the_library(ctx, bunny)
    test_tree_gpu.test_registration_real_scan_pair_against_bun_conf(ctx, bunny)
    test_tree_batch_gpu.test_registration_batch_is_bitwise_the_serial_registration(ctx, bunny)
    test_tree_batch_gpu.test_registration_batch_matches_reference_trace(ctx)
    test_tree_batch_gpu.test_registration_batch_ill_conditioned_pair_falls_back_like_the_serial_path(ctx, bunny)


CASES = {
    "estep_target_gbs": [(0, _flat), (6000, _flat)],
    "pace_start": [(5800, _flat)],
    "pace_forget": [(1, _flat)],
    "predict_single_row": [(1, _flat)],
    "tree_no_chol": [(1, _tree), (1, _fullcov)],
    "tree_rel": [(1, _tree)],
    "tree_ahead": [(0, _tree), (1, _tree), (4, _tree)],
    "tree_tickets": [(1, _tree)],
    "tree_overlap": [(0, _tree)],
    "fullcov_two_pass": [(1, _fullcov)],
    "kmpp_two_launches": [(1, _kmeans)],
    "kmeans_acc_regs": [(1, _kmeans)],
    "ipc_timeout_s": [(5, None)],            # behaviour: tests/test_multirank_gpu.py (a peer that never arrives)
    "reg_device_solve": [(1, _reg)],
}


def test_every_option_has_a_case(ctx):
    assert sorted(ctx.config_names()) == sorted(CASES)


@pytest.mark.parametrize("name,value,check", [(n, v, c) for n, cases in CASES.items() for v, c in cases])
def test_option_against_its_golden(ctx, bunny, name, value, check):
    before = ctx.config_get(name)
    with ctx.config(**{name: value}):
        assert ctx.config_get(name) == value
        if check is not None:
            check(ctx, bunny)
    assert ctx.config_get(name) == before


def test_options_start_from_the_environment_and_are_range_checked(monkeypatch):
    import hgmm_amd
    monkeypatch.setenv("HGMM_TREE_AHEAD", "5")
    monkeypatch.setenv("HGMM_TREE_OVERLAP", "0")
    c = hgmm_amd.Context(0)
    try:
        assert c.config_get("tree_ahead") == 5 and c.config_get("tree_overlap") == 0 and c.config_get("tree_tickets") == 0
        monkeypatch.setenv("HGMM_TREE_AHEAD", "7")                 # (not looked at again)
        assert c.config_get("tree_ahead") == 5
        with pytest.raises(hgmm_amd.HgmmError):
            c.config_set("tree_ahead", 1000)
        with pytest.raises(hgmm_amd.HgmmError):
            c.config_set("no_such_option", 1)
    finally:
        c.close()
