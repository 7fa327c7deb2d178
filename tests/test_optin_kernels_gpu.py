"""The opt-in kernel variants kept in the tree as measured experiments (DESIGN.md "learned" 9-11) stay CORRECT: each
runs tools/conv_probe.py --check (relative error vs torch's fp32 convolution) in a subprocess with its environment
switch, because the switches are read once per process."""
import os
import re
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = ["1,32,32,64,96,3", "2,16,16,128,64,3", "1,8,8,256,128,3", "1,8,8,128,256,1", "1,40,24,32,64,3", "1,16,16,64,64,1"]


@pytest.mark.parametrize("env,mode", [({"OSM_TALL_MINM": "1"}, "bf16x6"), ({"OSM_TALL_MINM": "1"}, "f16"),
                                      ({"OSM_SKINNY_MAXM": "1024"}, "bf16x6"), ({"OSM_SKINNY_MAXM": "1024"}, "f16"),
                                      ({"OSM_BRING16": "6", "OSM_BRING8": "9"}, "bf16x6")])
def test_optin_conv_variants_are_correct(env, mode):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    cmd = [sys.executable, os.path.join(ROOT, "tools", "conv_probe.py"), "--check", "--iters", "2", "--mode", mode]
    for s in SHAPES:
        cmd += ["--shape", s]
    out = subprocess.run(cmd, env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    errs = [float(m) for m in re.findall(r"relerr ([0-9.e+-]+)", out.stdout)]
    assert len(errs) == len(SHAPES), out.stdout
    tol = 2e-3 if mode == "f16" else 5e-6        # f16: operands and result rounded to half
    assert max(errs) < tol, out.stdout
