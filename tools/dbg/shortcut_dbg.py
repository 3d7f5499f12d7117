import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import numpy as np, oracle
from yolo_quantization_amd import binding
binding.init(0)
B, Cc, H, W = 1, 32, 16, 16
rng = np.random.default_rng(B + Cc + H)
a = rng.integers(0, 256, (B, Cc, H, W), dtype=np.uint8); b = rng.integers(0, 256, (B, Cc, H, W), dtype=np.uint8)
for (sa, za, sb, zb, so, zo) in ((6.6 / 255, 23, 6.6 / 255, 23, 9.0 / 255, 23), (6 / 255, 0, 16 / 255, 128, 7.5 / 255, 60), (0.2, 255, 0.3, 0, 0.0097, 10)):
    Ka = oracle.shortcut_multiplier(np.float32(sa), np.float32(so)); Kb = oracle.shortcut_multiplier(np.float32(sb), np.float32(so))
    ta = binding.DevTensor.from_nchw(a, za); tb = binding.DevTensor.from_nchw(b, zb); ty = binding.DevTensor(B, H, W, Cc, zo)
    binding.check(binding.shim().mi355_shortcut_forward(ta.ref(), tb.ref(), ty.ref(), Ka, Kb, za, zb, zo, None), "shortcut")
    want = oracle.shortcut_u8(a, b, Ka, Kb, za, zb, zo); got = ty.to_nchw()
    bad = np.argwhere(got != want)
    print(Ka, Kb, za, zb, zo, "mismatches", len(bad))
    for idx in bad[:8]:
        i = tuple(idx); print("  ", i, "a", a[i], "b", b[i], "got", got[i], "want", want[i])
