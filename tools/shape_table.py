"""Per-shape kernel table from `python bench.py --profile-shapes --dump-kernels F`: time against a simple floor
max(flops / 1.25 PFLOP/s, algorithmic bytes / 6 TB/s) -- ranks the shapes by the time they spend above that floor.
    shape_table.py A.json [B.json sa sb]   with B: per-denoise-step table = (B - A) / (sb - sa)  (the VAE / prior kernels cancel)"""
import json, sys
k = json.load(open(sys.argv[1]))
if len(sys.argv) > 4:
    kb = json.load(open(sys.argv[2])); d = float(sys.argv[4]) - float(sys.argv[3])
    k = {key: [(kb[key][0] - k.get(key, [0, 0, 0, 0])[0]) / d, (kb[key][1] - k.get(key, [0, 0, 0, 0])[1]) / d,
               (kb[key][2] - k.get(key, [0, 0, 0, 0])[2]) / d, (kb[key][3] - k.get(key, [0, 0, 0, 0])[3]) / d] for key in kb}
    k = {key: v for key, v in k.items() if v[0] > 0}
tot = sum(v[1] for v in k.values())
rows = []
for key, (n, sec, fl, by) in k.items():
    floor = max(fl / 1.25e15, by / 6e12)
    rows.append((sec - floor, key, n, sec, fl, by, floor))
rows.sort(reverse=True)
print(f"total kernel seconds {tot:.4f}")
print(f"{'excess ms':>9} {'ms':>8} {'%':>5} {'n':>6} {'ms/launch':>9} {'TF/s':>7} {'GB/s':>7} {'AI':>6}  key")
for ex, key, n, sec, fl, by, floor in rows[:60]:
    print(f"{ex * 1e3:9.2f} {sec * 1e3:8.2f} {100 * sec / tot:5.1f} {n:6.0f} {sec / n * 1e3:9.3f} {fl / sec / 1e12:7.1f} {by / sec / 1e9:7.0f} {fl / max(by, 1):6.0f}  {key}")
