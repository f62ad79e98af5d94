"""Does the pass time depend on where the tensors sit?  (k_forward measured 1.37 ms in some processes and 1.53 in
others on the same box.)  One process, E / out / workspace carved out of big buffers at different byte offsets."""
import sys, os, time, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hmm_layer_amd import engine
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _model import gene15
dev = torch.device('cuda:0')
b, L, q = 1024, 100000, 15
A, pi = gene15(dev)
A = A.contiguous(); pi = pi.reshape(1, q).contiguous()
n = b * L * q
SLACK = 1 << 22                                   # floats
Ebig = torch.rand(n + SLACK, device=dev) * 0.9 + 0.05
Obig = torch.empty(n + SLACK, device=dev)
lib = engine.lib()
need = lib.hmm_workspace_bytes(engine.OP_POSTERIOR, 1, b, L, q)
Wbig = torch.empty(need + (1 << 24), dtype=torch.uint8, device=dev)
ll = torch.empty((1, b), dtype=torch.float64, device=dev)
st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
print("base addresses mod 2^21: E %x out %x ws %x" % (Ebig.data_ptr() % (1 << 21), Obig.data_ptr() % (1 << 21), Wbig.data_ptr() % (1 << 21)))

def run(eo, oo, wo):
    Ep = Ebig.data_ptr() + 4 * eo; Op = Obig.data_ptr() + 4 * oo; Wp = Wbig.data_ptr() + wo
    def call():
        rc = lib.hmm_posterior(A.data_ptr(), pi.data_ptr(), Ep, 1, b, L, q, ctypes.c_float(engine.EPS), engine.POST_PROB, Op, ll.data_ptr(), Wp, need, st)
        assert rc == 0, rc
    call(); call()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): call()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 5 * 1e3

base = run(0, 0, 0)
print("offsets 0/0/0: %.3f ms" % base)
for wo in (256, 1024, 4096, 16384, 65536, 1 << 18, 1 << 20, 3 << 20, (1 << 21) + 4096):
    print("ws +%8d B: %.3f ms" % (wo, run(0, 0, wo)), flush=True)
for eo in (64, 256, 1024, 4096, 1 << 16, 1 << 19, 1 << 20):
    print("E  +%8d B: %.3f ms" % (4 * eo, run(eo, 0, 0)), flush=True)
for oo in (64, 1024, 1 << 16, 1 << 20):
    print("out+%8d B: %.3f ms" % (4 * oo, run(0, oo, 0)), flush=True)
