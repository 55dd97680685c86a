"""Diagnostic (GPU box, stamps build): cadence of the 80x64 GEMM launch geometry (256 workgroups x 512 threads, 144 KiB of
LDS) with an empty kernel body vs the real kernel, 50 launches per graph."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
subprocess.check_call([sys.executable, "-m", "ladiff_amd.build", "--stamps"], cwd=ROOT, stdout=subprocess.DEVNULL)
import torch
L = ctypes.CDLL(os.path.join(ROOT, "ladiff_amd", "libladiff_hip_stamps.so"))
dev = "cuda:0"; M, N, K = 1280, 1024, 256
A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.05; b = torch.randn(N, device=dev)
Y = torch.empty(M, N, device=dev)
s = torch.cuda.Stream()
def timeit(reps=50):
    fn = lambda: L.ladiff_gemm_resident(ctypes.c_void_p(A.data_ptr()), K, None, 0, K, ctypes.c_void_p(W.data_ptr()), K, ctypes.c_void_p(b.data_ptr()),
                                        None, 0, ctypes.c_void_p(Y.data_ptr()), N, M, N, K, 0, 1, None, ctypes.c_void_p(s.cuda_stream))
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps): fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(10): g.replay()
        e1.record(s); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (10 * reps)
L.ladiff_debug_set_stamps(ctypes.c_void_p(0))
print(f"real kernel      : {timeit():.2f} us / launch")
L.ladiff_debug_set_stamps(ctypes.c_void_p(1))
print(f"empty body       : {timeit():.2f} us / launch")
