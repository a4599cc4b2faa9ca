"""Host logic of the AnyMDP boundary (CPU only): task validation as in AnyMDPEnv.set_task, table layout."""
import numpy as np
import pytest

from xenoverse_amd.anymdp.tables import build_tables, from_blocked, row_cdf, row_lines, to_blocked, validate_task
from util import golden_files, load_anymdp_golden


def _task():
    return load_anymdp_golden(golden_files("anymdp_16x4")[0])[1]


def test_tables_shapes_and_padding():
    t0 = _task()
    t1 = load_anymdp_golden(golden_files("anymdp_16x4")[1])[1]
    tab = build_tables([t0, t1])
    assert tab["cdf"].shape == (2, 16, 4, 16) and tab["rs"].shape == (2, 16, 4, 16, 2)
    assert tab["cdf"].dtype == np.float64 and tab["rs"].dtype == np.float32
    assert tab["max_steps"][0] == int(np.ceil(t0["max_steps"]))
    assert np.all(tab["cdf"][..., -1] == 1.0)
    for s in t0["s_e"]:
        assert (int(tab["term_mask"][0, 0]) >> int(s)) & 1
    assert bin(int(tab["term_mask"][0, 0])).count("1") == len(t0["s_e"])


def test_row_cdf_is_numpy_choice_cdf():
    p = np.random.RandomState(0).dirichlet(np.ones(16), size=(3, 4))
    c = row_cdf(p)
    ref = np.cumsum(p, -1)
    ref /= ref[..., -1:]
    assert np.array_equal(c, ref)
    z = row_cdf(np.zeros((2, 5)))
    assert np.all(z == 1.0)


def test_validate_rejects_bad_rows_like_reference():
    t = _task()
    bad = dict(t)
    T = t["transition"].copy()
    s_ok = [s for s in range(T.shape[0]) if s not in set(t["s_e"])][0]
    T[s_ok, 0] *= 0.5
    bad["transition"] = T
    with pytest.raises(Exception, match="Transition Matrix Sum != 1"):
        validate_task(bad)
    bad2 = dict(t)
    bad2["s_0"] = np.array([int(t["s_e"][0])])
    bad2["s_0_prob"] = np.array([1.0])
    with pytest.raises(Exception):
        validate_task(bad2)
    bad3 = dict(t)
    bad3["task_type"] = "FOO"
    with pytest.raises(NotImplementedError):
        validate_task(bad3)


def test_mixed_action_spaces_rejected():
    t = _task()
    t2 = dict(t)
    t2["na"] = 5
    with pytest.raises(ValueError):
        build_tables([t, t2], validate=False)


@pytest.mark.parametrize("S", [8, 16, 20, 64, 100, 256])
def test_blocked_row_layout(S):
    """include/xeno.h "rows": fence line + blocks of 7 entries {cdf, reward, noise} + 16 bytes of metadata"""
    rng = np.random.RandomState(S)
    cdf = np.sort(rng.random_sample((2, 3, S)), axis=-1)
    rs = rng.standard_normal((2, 3, S, 2)).astype(np.float32)
    rows = to_blocked(cdf, rs)
    from xenoverse_amd.anymdp.tables import row_blocks
    NB = row_blocks(S)        # ceil(S/7) rounded up to a multiple of G = ceil(ceil(S/7)/16)
    assert NB >= (S + 6) // 7 and NB % ((((S + 6) // 7) + 15) // 16) == 0 and row_lines(S) == 1 + NB
    assert rows.shape == (2, 3, 1 + NB, 16) and rows.dtype == np.float64
    raw = rows.view(np.uint8).reshape(2, 3, 1 + NB, 128)
    for j in sorted({0, 1, min(6, S - 1), S // 2, S - 1}):   # entry j: block j // 7, slot j % 7, 16 bytes
        ent = raw[:, :, 1 + j // 7, 16 * (j % 7):16 * (j % 7) + 16]
        assert np.array_equal(np.ascontiguousarray(ent[..., :8]).view(np.float64)[..., 0], cdf[..., j])
        assert np.array_equal(np.ascontiguousarray(ent[..., 8:]).view(np.float32), rs[..., j, :])
    assert np.all(raw[:, :, 0, :] == 0) and np.all(raw[:, :, 1:, 112:] == 0)   # fence / metadata: the engine's
    ent_all = np.ascontiguousarray(raw[:, :, 1:, :112]).view(np.float64).reshape(2, 3, NB * 7, 2)
    assert np.all(ent_all[:, :, S:, 0] == 2.0) and np.all(ent_all[:, :, S:, 1] == 0.0)   # padding never compares <= u
    c2, r2 = from_blocked(rows, S)
    assert np.array_equal(c2, cdf) and np.array_equal(r2, rs)


def test_bandit_task_is_embedded_in_two_states():
    """state_space = 1 (multi-armed bandit): the reference terminates every step; the tables hold the state and an
    absorbing terminal copy of it"""
    from xenoverse_amd.anymdp import AnyMDPTaskSampler
    t = AnyMDPTaskSampler(1, 5, seed=0)
    tab = build_tables([t])
    assert tab["S"] == 2 and tab["A"] == 5
    assert np.array_equal(tab["cdf"][0, 0], np.tile([0.0, 1.0], (5, 1)))
    assert np.array_equal(tab["rs"][0, 0, :, 1, 0], t["reward"][0, :, 0].astype(np.float32))
    assert int(tab["term_mask"][0, 0]) == 2 and tab["state_map"][0, 0] == tab["state_map"][0, 1]
    assert tab["s0_ids"][0, 0] == 0 and tab["max_steps"][0] == 1
