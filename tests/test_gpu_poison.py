"""The engine's results must not depend on what its device workspaces held before: fresh allocations are zero,
recycled ones are not (an intermittent failure of round 4 came from exactly that: the FP64 filter bank read the
stale rest of a row behind a short session's samples).  PEAQ_AMD_POISON=1 makes every workspace start as NaNs; a
child process runs ragged batches, a session fed in small pieces and a broker that way, and must print what the
parent computes without it."""
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

import cases as case_defs

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent

CHILD = r"""
import json, sys
sys.path.insert(0, %(tests)r); sys.path.insert(0, %(root)r)
import numpy as np
import gpu_common as gpu, cases as case_defs, gstpeaq_amd
out = {}
specs = [dict(kind="synth", seed=21, channels=2, n=90000), dict(kind="synth", seed=22, channels=2, n=31000, test_trim=700),
         dict(kind="synth", seed=23, channels=2, n=2500), dict(kind="synth", seed=24, channels=2, n=150000)]
pairs = [case_defs.make_inputs(s) for s in specs]
for adv in (0, 1):
    out["batch%%d" %% adv] = [[r["odg"]] + list(map(float, r["movs"])) for r in gpu.run_batch(pairs, adv, 2)]
    s = gstpeaq_amd.Session(gpu.ctx(), adv, 2)
    ref, test = pairs[1]
    for lo in range(0, len(ref), 500):
        s.push_ref(ref[lo:lo + 500]); s.push_test(test[lo:lo + 500])
    s.flush(); r = s.results(); s.close()
    out["session%%d" %% adv] = [r["odg"]] + list(map(float, r["movs"]))
    b = gstpeaq_amd.Broker(gpu.ctx(), 2, max_sessions=3, advanced=bool(adv))
    got = []
    for ref, test in pairs[:3]:                      # different lengths in the same launches
        got.append(b.open())
    for lo in range(0, 90000, 4096):
        for sid, (ref, test) in zip(got, pairs[:3]):
            if lo < len(ref): b.push(sid, 0, ref[lo:lo + 4096])
            if lo < len(test): b.push(sid, 1, test[lo:lo + 4096])
        b.tick()
    res = []
    for sid in got:
        b.flush(sid); b.tick()
        r = b.results(sid); res.append([r["odg"]] + list(map(float, r["movs"]))); b.close_session(sid)
    b.close()
    out["broker%%d" %% adv] = res
print("RESULT " + json.dumps(out))
"""


def _run(poison):
    env = dict(os.environ)
    env.pop("PEAQ_AMD_POISON", None)
    if poison:
        env["PEAQ_AMD_POISON"] = "1"
    code = CHILD % dict(tests=str(ROOT / "tests"), root=str(ROOT))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


def test_results_do_not_depend_on_what_the_workspaces_held():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU (there is no CPU fallback in the product)")
    clean, dirty = _run(False), _run(True)
    assert clean.keys() == dirty.keys()
    for k in clean:
        a, b = np.array(clean[k], dtype=float), np.array(dirty[k], dtype=float)
        assert a.shape == b.shape, k
        assert not np.isnan(b[..., 0]).any() or np.array_equal(np.isnan(a), np.isnan(b)), (k, b)
        # bit for bit, both versions (the default engine's sums have one owner and one order each)
        np.testing.assert_allclose(b, a, rtol=0, atol=0, equal_nan=True, err_msg=k)
