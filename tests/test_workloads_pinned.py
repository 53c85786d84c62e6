"""The synthetic benchmark workloads are committed data with pinned digests:
a numpy upgrade cannot change leapfrogs_per_launch, the tuned step sizes or the key under which
profiles/pmc_traffic.json files a launch's counters."""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from exmc_amd import models  # noqa: E402


def test_every_committed_array_matches_its_digest():
    z = np.load(models.WORKLOADS)
    assert set(z.files) == set(models.WORKLOAD_SHA256)
    for k, want in models.WORKLOAD_SHA256.items():
        assert hashlib.sha256(np.ascontiguousarray(z[k]).tobytes()).hexdigest() == want, k


def test_models_use_the_committed_arrays():
    import bench
    r = np.array(bench.synthetic_sv_returns())
    assert np.array_equal(r, models.workload("sv_returns")) and r.shape == (100,)
    # the series the golden oracle traces were made from is the same one
    g = np.load(os.path.join(ROOT, "tests", "golden", "oracle_traces.npz"))
    assert np.array_equal(g["sv_returns"], r)
    X, y = models.logistic_data()
    assert X.shape == (500, 20) and y.shape == (500,) and set(np.unique(y)) == {0.0, 1.0}
    lg = models.logistic()
    assert np.array_equal(lg.data[:10000].reshape(500, 20), X) and np.array_equal(lg.data[10000:], y)
    u, start, floor, yy = models.radon_data()
    assert (len(u), len(start), len(floor), len(yy)) == (85, 86, 919, 919) and start[-1] == 919
    rd = models.radon()
    assert rd.d == 90 and rd.data.size == 85 + 86 + 919 + 919
    # the arrays handed out are copies: a caller cannot edit the pinned workload
    X[0, 0] = 1e9
    assert models.logistic_data()[0][0, 0] != 1e9


def test_a_tampered_file_is_refused(tmp_path, monkeypatch):
    z = dict(np.load(models.WORKLOADS))
    z["sv_returns"] = z["sv_returns"] + 1e-12
    p = tmp_path / "workloads.npz"
    np.savez(p, **z)
    monkeypatch.setattr(models, "WORKLOADS", str(p))
    monkeypatch.setattr(models, "_workloads", None)
    try:
        models.workload("sv_returns")
    except ValueError as e:
        assert "pinned digest" in str(e)
    else:
        raise AssertionError("a modified workload file was accepted")
    monkeypatch.setattr(models, "_workloads", None)


def test_other_arguments_still_generate():
    X, y = models.logistic_data(seed=7, n=50, k=3)
    assert X.shape == (50, 3) and y.shape == (50,)
    assert len(models.sv_returns(seed=1)) == 100
