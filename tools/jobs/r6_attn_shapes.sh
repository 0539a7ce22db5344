# Round 6: block shapes (eight waves x 16 queries against four x 32) of the VAE mid attention (d = 512) and of the cross attention to the 77 text tokens (d = 80 / 160), lab build, interleaved
O=gpurun_out/r6_attn_shapes; mkdir -p $O
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $O/ab.txt
import os, sys, subprocess
code = r"""
import os, sys, torch
sys.path.insert(0, '.')
from videovanish_amd import hip
hip._LIB_PATH = 'videovanish_amd/csrc/ab/attnshapes.so'
DT = hip.F16; td = torch.float16
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
kind = sys.argv[1]
if kind == 'vae':      # VAE mid attention: 4 frames x 14400 tokens, one head of 512
    B, N, D = 4, 14400, 512
    q = torch.randn(B, N, 3 * D, device='cuda').to(td) * 0.2
    out = torch.empty(B * N, D, dtype=td, device='cuda')
    fn = lambda: hip.attention(DT, q, q, q, out, B=B, heads=1, Nq=N, Nkv=N, D=D, q_bs=N * 3 * D, k_bs=N * 3 * D, v_bs=N * 3 * D, o_bs=N * D, q_rs=3 * D, k_rs=3 * D, v_rs=3 * D, o_rs=D, k_off=D, v_off=2 * D)
else:                  # cross attention: 32 frames x N tokens x 8 heads against 77 text tokens
    D = int(kind); N = 3600 if D == 80 else 920; C = 8 * D; B = 32
    q = torch.randn(B * N, C, device='cuda').to(td) * 0.3
    kv = torch.randn(77, 2 * C, device='cuda').to(td) * 0.3
    out = torch.empty(B * N, C, dtype=td, device='cuda')
    fn = lambda: hip.attention(DT, q, kv, kv, out, B=B, heads=8, Nq=N, Nkv=77, D=D, q_bs=N * C, k_bs=0, v_bs=0, o_bs=N * C, q_rs=C, k_rs=2 * C, v_rs=2 * C, o_rs=C, v_off=C)
fn(); torch.cuda.synchronize()
ref = out.clone()
print(f"{kind}: {timeit(fn):.4f} ms  checksum {float(ref.float().abs().sum()):.6e}")
"""
open('/tmp/attn_shape_probe.py', 'w').write(code)
for r in range(3):
    for kind, var in (('vae', 'VV_ATTN512_FORM'), ('80', 'VV_ATTNX_FORM'), ('160', 'VV_ATTNX_FORM')):
        for form in ('0', '1'):
            env = dict(os.environ, **{var: form})
            o = subprocess.run([sys.executable, '/tmp/attn_shape_probe.py', kind], env=env, capture_output=True, text=True)
            print(f"round {r} {var}={form}", (o.stdout.strip().splitlines() or [o.stderr[-300:]])[-1])
PY
