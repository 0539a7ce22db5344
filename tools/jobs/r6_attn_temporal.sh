# Round 6: the temporal attention core (32 frames, batch = pixel) on two waves x 16 queries per block instead of one wave x 32 (lab build ab/attnt.so, VV_ATTNT_FORM=1), interleaved
O=gpurun_out/r6_attn_temporal; mkdir -p $O
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $O/ab.txt
import os, sys, subprocess
code = r"""
import os, sys, torch
sys.path.insert(0, '.')
from videovanish_amd import hip
hip._LIB_PATH = 'videovanish_amd/csrc/ab/attnt.so'
DT = hip.F16; td = torch.float16
C = int(sys.argv[1]); HW = int(sys.argv[2]); Fr = 32
qkv = (torch.randn(Fr * HW, 3 * C, device='cuda') * 0.3).to(td)
o = torch.empty(Fr * HW, C, dtype=td, device='cuda')
fn = lambda: hip.attention(DT, qkv, qkv, qkv, o, B=HW, heads=8, Nq=Fr, Nkv=Fr, D=C // 8, q_bs=3 * C, k_bs=3 * C, v_bs=3 * C, o_bs=C, q_rs=HW * 3 * C, k_rs=HW * 3 * C, v_rs=HW * 3 * C, o_rs=HW * C, k_off=C, v_off=2 * C)
for _ in range(5): fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(40): fn()
e1.record(); torch.cuda.synchronize()
print(f"C={C} HW={HW}: {e0.elapsed_time(e1) / 40:.4f} ms  sum {float(o.float().sum()):.4e}")
"""
open('/tmp/attn_t_probe.py', 'w').write(code)
for r in range(3):
    for C, HW in ((640, 3600), (1280, 920)):
        for form in ('0', '1'):
            o = subprocess.run([sys.executable, '/tmp/attn_t_probe.py', str(C), str(HW)], env=dict(os.environ, VV_ATTNT_FORM=form), capture_output=True, text=True)
            print(f"round {r} form {form}", (o.stdout.strip().splitlines() or [o.stderr[-300:]])[-1])
PY
