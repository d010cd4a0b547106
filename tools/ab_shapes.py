import sys, json, subprocess, os
# A/B of two library builds on the SAME box (boxes differ by ~8 % on the roofline shapes): copy the two builds to
# tools/lib/libalignq_<name>.so, then  python tools/ab_shapes.py base new
for rep in range(2):
    for name in sys.argv[1:]:
        env = dict(os.environ, ALIGNQ_AB_SO=f"tools/lib/libalignq_{name}.so")
        out = subprocess.run([sys.executable, "-c", "import os,sys; sys.path.insert(0,'.'); from alignq_amd import _lib as L; L.SO_PATH=os.environ['ALIGNQ_AB_SO']; import runpy; sys.argv=['tools/roofline_shapes.py']; runpy.run_path('tools/roofline_shapes.py', run_name='__main__')"],
                             env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
        d = json.loads(out)
        print(name, {k: (round(v["fwd_us"], 1), round(v["bwd_us"], 1)) for k, v in d.items() if k.startswith("site")})
