"""Diagnostic (GPU box): where a K-resident GEMM workgroup spends its cycles (s_memtime stamps, separate build)."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
subprocess.check_call([sys.executable, "-m", "ladiff_amd.build", "--stamps"], cwd=ROOT, stdout=subprocess.DEVNULL)
import torch
L = ctypes.CDLL(os.path.join(ROOT, "ladiff_amd", "libladiff_hip_stamps.so"))
dev = "cuda:0"; M = 1280
for (N, K, ln, SPLIT) in [(1024, 256, 0, 0), (1024, 256, 0, 1), (768, 256, 0, 0), (768, 256, 0, 1), (256, 256, 0, 1), (256, 1024, 0, 1)]:
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.05; b = torch.randn(N, device=dev)
    Y = torch.empty(4, M, N, device=dev); res = torch.randn(M, N, device=dev)
    st = torch.zeros(4096, 8, dtype=torch.int64, device=dev)
    g_ = torch.ones(256, device=dev); be = torch.zeros(256, device=dev); xo = torch.empty(M, 256, device=dev)
    P = lambda t: ctypes.c_void_p(t.data_ptr()) if ln else None
    L.ladiff_debug_set_stamps(ctypes.c_void_p(st.data_ptr()))
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(5):
            L.ladiff_gemm_resident(ctypes.c_void_p(A.data_ptr()), K, None, 0, K, ctypes.c_void_p(W.data_ptr()), K, ctypes.c_void_p(b.data_ptr()),
                                   ctypes.c_void_p(res.data_ptr()), N, ctypes.c_void_p(Y.data_ptr()), N, M, N, K, 0, SPLIT, None, ctypes.c_void_p(s.cuda_stream))
        torch.cuda.synchronize()
    t = st.cpu().double()
    nb = int((t[:, 0] > 0).sum())
    t = t[:nb]
    d = (t[:, 1:8] - t[:, 0:1])
    names = ["issued", "A landed", "LN done", "last sub starts", "mfma done", "stored", "C staged"] if ln else ["issued", "sub0 landed", "sub0 done", "last sub starts", "mfma done", "stored", "C staged"]
    t0 = t[:, 0].min()
    print(f"N={N} K={K} split={SPLIT}: {nb} workgroups; start spread {float((t[:,0]-t0).max()):.0f} cyc; last store issued {float(t[:,6].max()-t0):.0f} cyc after the first workgroup started")
    print("   median cycles since workgroup start: " + ", ".join(f"{n} {d[:, i].median():.0f}" for i, n in enumerate(names)))
