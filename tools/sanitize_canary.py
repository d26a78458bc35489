"""tools/sanitize.sh's canary: a deliberate defect the sanitizer of this run must report, through the same path the tests take
(python with the runtime preloaded -> ctypes -> the instrumented libgd3d.so).  asan: gd3d_loss_fused_cpu writes 300 gradient rows
into a malloc'ed array of 299.  tsan: two threads run the twin on the SAME output arrays at the same time."""
import ctypes, os, sys, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mmdet3d_gaussian_amd as amd
lib = amd.load_library()
kind = sys.argv[1]
n = 300
rng = np.random.default_rng(0)
t = (rng.random((n, 7)) + 0.5).astype(np.float32)
p = (t + 0.05 * rng.standard_normal((n, 7))).astype(np.float32)
prm = amd.make_params('kld3d', 'log1p', 1.0, 1.0, (0, 0, 0.5), {})
libc = ctypes.CDLL(None)
libc.malloc.restype = ctypes.c_void_p
libc.malloc.argtypes = [ctypes.c_size_t]
vp = lambda a: ctypes.c_void_p(a.ctypes.data)
if kind == 'asan':
    gp = libc.malloc((n - 1) * 28)          # one row short
    rc = lib.gd3d_loss_fused_cpu(ctypes.byref(prm), vp(p), vp(t), None, None, n, 1.0, None, None, ctypes.c_void_p(gp), None, None, 1)
    print('canary call returned', rc)
else:
    gp = np.empty((n, 7), np.float32); loss = np.empty(n, np.float32)
    def work():
        for _ in range(200):
            lib.gd3d_loss_fused_cpu(ctypes.byref(prm), vp(p), vp(t), None, None, n, 1.0, vp(loss), None, vp(gp), None, None, 1)
    th = [threading.Thread(target=work) for _ in range(2)]
    [x.start() for x in th]; [x.join() for x in th]
    print('canary threads done')
