"""The oracle's run-time version of the reference's settings.h switches (settings.h:47-97) against the
REAL reference compiled with each switch flipped (oracle/Makefile ref_variants: the reference's own
sources, settings.h included through a generated wrapper that re-defines one macro; fixtures by
tools/make_golden.py settings -> tests/golden/ref_e2e_settings.json).  CPU only."""
import json

import numpy as np
import pytest

import cases as case_defs
import oracle_lib as orc


def _records(golden_dir):
    return json.loads((golden_dir / "ref_e2e_settings.json").read_text())


def test_fixture_covers_every_switch_and_every_switch_matters(golden_dir):
    recs = _records(golden_dir)
    by_variant = {}
    for r in recs:
        by_variant.setdefault(r["variant"], []).append(r)
    assert sorted(by_variant) == ["center1", "clamp1", "dcafter", "floor1", "swapmod0", "swapslope1"]
    for variant, rs in by_variant.items():
        (field, value), = rs[0]["settings"].items()
        assert field in orc.SETTINGS_FIELDS and value != orc.SETTINGS_DEFAULT[field], variant
        # the flipped switch changes the result of at least one case (else the fixture pins nothing)
        assert any(r["movs_changed_vs_default"] or r["odg"] != r["odg_default"] for r in rs), variant


@pytest.mark.parametrize("advanced", [0, 1])
def test_oracle_matches_the_reference_built_with_other_settings(golden_dir, advanced):
    worst = 0.0
    try:
        for rec in _records(golden_dir):
            case = rec["case"]
            if case["advanced"] != advanced:
                continue
            ref, test = case_defs.make_inputs(case)
            orc.set_settings(**rec["settings"])
            got = orc.run_pair(advanced, ref, test)
            exp = np.array([float(v) for v in rec["movs"]])
            np.testing.assert_allclose(got["movs"][: len(exp)], exp, rtol=1e-11, atol=1e-13,
                                       err_msg=f"{rec['variant']} {case['name']}")
            assert abs(got["odg"] - rec["odg"]) < 1e-11 and abs(got["di"] - rec["di"]) < 1e-11
            worst = max(worst, abs(got["odg"] - rec["odg"]))
    finally:
        orc.set_settings()
    print(f"oracle vs reference variants, advanced={advanced}: max |dODG| {worst:.2e}")
