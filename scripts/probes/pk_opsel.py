"""Driver of scripts/probes/pk_opsel.hip (run on the GPU box): every victim form beside every co-runner kind, two streams.
Prints in how many of the victim launches (and lanes) the exactly known result is wrong."""
import ctypes as C
import os

import numpy as np
import torch

lib = C.CDLL(os.path.abspath("variants/libpk_opsel.so"))
lib.victim.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
lib.corunner.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
import sys
SHORT = len(sys.argv) > 1 and sys.argv[1] == "short"
N, BLOCKS, LAUNCHES = 4096, 64, 20
print("co-runner as", "300 short kernels" if SHORT else "one long kernel")
t = torch.arange(256, dtype=torch.float32, device="cuda")
x, y = t + 1, 2 * t + 1
FORMS = [(0, "plain", (N * x, N * y)),
         (1, "op_sel:[0,1] op_sel_hi:[1,0]  (src1 halves swapped)", (N * y, N * x)),
         (2, "op_sel_hi:[1,0]               (src1.lo to both)", (N * x, N * x)),
         (3, "op_sel:[1,0] op_sel_hi:[0,1]  (src0 halves swapped)", (N / 2 * (x + y), N / 2 * (x + y))),
         (4, "op_sel:[0,1]                  (src1.hi to both)", (N * y, N * y)),
         (5, "v_pk_fma_f32, src1 halves swapped", (N * y, N * x)),
         (6, "v_pk_fma_f32, plain", (N * x, N * y))]
KINDS = [(0, "v_mfma_f32_32x32x16_f16", 60000), (3, "v_mfma_f32_16x16x32_f16", 120000),
         (4, "v_mfma_f32_32x32x16_bf16", 60000), (5, "v_mfma_f32_32x32x8_f16", 60000), (1, "v_mfma_f32_32x32x2_f32", 30000)]
FORMS_RUN = (0, 1, 2, 3, 4, 5, 6)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
sink = torch.zeros(4, device="cuda")
for kind, kname, kiters in KINDS:
    for form, fname, (wx, wy) in [f for f in FORMS if f[0] in FORMS_RUN]:
        want = torch.stack([wx, wy], 1).repeat(BLOCKS, 1)                      # [BLOCKS*256, 2]
        outs = [torch.empty((BLOCKS * 256, 2), device="cuda") for _ in range(LAUNCHES)]
        torch.cuda.synchronize()
        if kind >= 0:
            if SHORT:       # many short co-runner kernels: waves start and end all the time
                for _ in range(300):
                    assert lib.corunner(kind, sink.data_ptr(), 512, max(kiters // 300, 50), C.c_void_p(s2.cuda_stream)) == 0
            else:
                assert lib.corunner(kind, sink.data_ptr(), 512, kiters, C.c_void_p(s2.cuda_stream)) == 0
        for o in outs:
            assert lib.victim(form, o.data_ptr(), BLOCKS, N, C.c_void_p(s1.cuda_stream)) == 0
        s1.synchronize()
        still = not s2.query()                                                 # the co-runner outlived the victims
        torch.cuda.synchronize()
        bad_l = sum(1 for o in outs if not torch.equal(o, want))
        lanes = np.zeros(64, np.int64)
        for o in outs:
            ne = (o != want).any(1).reshape(BLOCKS, 4, 64).sum((0, 1)).cpu().numpy()
            lanes += ne
        msg = "co-runner %-26s victim %-52s: %2d / %d launches wrong" % (kname, fname, bad_l, LAUNCHES)
        if bad_l:
            msg += ", wrong results by lane quarter %s" % [int(lanes[q * 16:(q + 1) * 16].sum()) for q in range(4)]
        if kind >= 0 and not still:
            msg += "   (co-runner ended first)"
        print(msg)
