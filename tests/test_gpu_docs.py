"""The ctypes stub INTEGRATION.md shows a maintainer of the reference is executed verbatim: it must run
and reproduce the upstream known-answer vector."""
import os
import re

import numpy as np
import pytest
import torch

from cases import KAT1

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_integration_md_ctypes_stub_runs_as_written():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    code = re.search(r"```python\n(import ctypes, torch\n.*?)```", text, re.S).group(1)
    code = code.replace('"rnnt_amd/lib/librnnt_mi355.so"', repr(os.path.join(ROOT, "rnnt_amd", "lib", "librnnt_mi355.so")))
    dev = torch.device("cuda:0")
    ns = dict(acts=torch.from_numpy(KAT1["acts"]).to(dev), labels=torch.from_numpy(KAT1["labels"]).to(dev),
              act_lens=torch.from_numpy(KAT1["act_lens"]).to(dev), label_lens=torch.from_numpy(KAT1["label_lens"]).to(dev),
              grad_out=torch.ones(1, device=dev))
    ns["grads"] = torch.empty_like(ns["acts"])
    exec(code, ns)
    torch.cuda.synchronize()
    assert ns["st"] == 0
    assert abs(ns["loss"].item() - KAT1["cost"]) <= 1e-5 * KAT1["cost"]
    assert np.abs(ns["grads"].cpu().numpy() - KAT1["grads"]).max() <= 1e-6
