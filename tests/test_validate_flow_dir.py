"""scripts/validate_flow_dir.py -- the one-command check for someone with a real flow directory -- on a tiny synthetic directory:
the CPU engines here (oracle, and the unmodified reference Python where /root/reference exists), the product path on the GPU box."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import psfm_synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _write_dir(tmp_path, T=7, H=40, W=56, seed=8):
    from point_trajectory.utils import write_flo
    d = psfm_synth.synth_sequence(T, H, W, seed=seed, sigma=0.3, n_occluders=1, stride2=True)
    for sub, key in (("flow_f", "flows_f"), ("flow_b", "flows_b"), ("flow_f2", "flows_f2"), ("flow_b2", "flows_b2")):
        os.makedirs(tmp_path / sub)
        for i, f in enumerate(d[key]):
            write_flo(str(tmp_path / sub / ("%05d.flo" % i)), f)
    return str(tmp_path)


def _run(args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "validate_flow_dir.py")] + args, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    return json.loads(r.stdout)


@pytest.mark.parametrize("optimize", [False, True])
def test_cpu_engines_on_a_synthetic_directory(tmp_path, optimize):
    rec = _run([_write_dir(tmp_path), "--no-gpu", "--ref-frames", "4"] + (["--optimize"] if optimize else []))
    assert rec["ok"] and rec["frames"] == 7 and rec["engines"]["oracle"]["points"] > 0
    from oracle import ref_shim
    if ref_shim.available():        # the build container: the unmodified reference ran on the first pairs and agrees
        key = [k for k in rec["parity"] if k.startswith("reference_vs_oracle")][0]
        assert rec["parity"][key]["ok"] and rec["parity"]["occlusion_maps_equal_reference_oracle"]
        assert rec["cpu_baseline"]["kind"] == "reference" and rec["cpu_baseline"]["value"] > 0
    else:
        assert rec["cpu_baseline"]["kind"] == "port"


@pytest.mark.gpu
@pytest.mark.parametrize("optimize", [False, True])
def test_product_path_on_a_synthetic_directory(tmp_path, optimize):
    rec = _run([_write_dir(tmp_path, T=9, H=60, W=84), "--ref-frames", "4"] + (["--optimize"] if optimize else []))
    p = rec["parity"]["hip_vs_oracle"]
    assert rec["ok"] and p["ids_lengths_equal"] and p["max_abs_dxy_px"] <= 1e-4
    if optimize:
        assert p["solve_iterations_equal"] and p["solve_terminations_equal"] and p["solves"] == 7
    else:
        assert p["max_abs_dxy_px"] == 0.0
    assert rec["engines"]["hip"]["points"] == rec["engines"]["oracle"]["points"] and rec["gpu_over_cpu_baseline"] > 0
