"""INTEGRATION.md's overlay executed against the real reference tree (build container only: needs /root/reference)."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.skipif(not os.path.isdir("/root/reference/model"), reason="the reference tree only exists in the build container")
def test_overlay_imports_callers_and_keeps_every_signature_and_attribute():
    """Every replaced reference module loaded from its own source + overlay.apply(): the unchanged callers (AnimalModel, MagicPony,
    Fauna, Ponymation, the predictors, misc, material, texture) import; every replaced function / class keeps the reference's
    signature as a prefix; every attribute the callers use exists, and the hot-path ones are served by this package."""
    p = subprocess.run([sys.executable, os.path.join(HERE, "overlay_check.py")], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    rep = json.loads(p.stdout.strip().splitlines()[-1])
    assert rep["import_errors"] == {}, rep["import_errors"]
    assert rep["signature_problems"] == [], rep["signature_problems"]
    assert rep["missing_attributes"] == [], rep["missing_attributes"]
    assert {"model.models.AnimalModel", "model.models.Fauna", "model.models.Ponymation", "model.predictors.InstancePredictorBase"} <= set(rep["imported"])
    assert len(rep["overlay_applied"]) == 9 and rep["dmtet_overlaid"] and rep["networks_in_use"].startswith("model.networks")
    # what the reference keeps serving after the additive overlay: image I/O helpers only (nothing on the hot path)
    assert set(rep["served_by_reference"]) <= {"model.render.util.load_image", "model.render.util.rgb_to_srgb", "model.render.util.save_image",
                                               "model.render.util.srgb_to_rgb", "model.render.util.save_image_raw"}, rep["served_by_reference"]
