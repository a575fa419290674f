"""One mesh under the fp32 policy (fused launches) for a kernel trace: rocprofv3 --kernel-trace --stats -- python scripts/prof_fp32_mesh.py"""
import os, sys
import numpy as np
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from meshanything_amd.config import MAConfig, DTYPE_F32
from meshanything_amd.engine import Engine
from meshanything_amd.checkpoint import synthetic_state_dict
cfg = MAConfig.full(dtype=DTYPE_F32, n_max_faces=800, max_batch=1)
eng = Engine(cfg)
eng.load_weights(synthetic_state_dict(cfg, init="diverse").items())
x = torch.from_numpy(np.load(os.path.join(REPO, "tests", "golden", "dataset.npz"))["mouse_norm"])[None].cuda()
o = eng.forward(x, suppress_eos=True)
torch.cuda.synchronize()
print(o["tokens"].shape)
