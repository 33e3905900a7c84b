import importlib, os, sys, subprocess, tempfile, re
ROOT="/root/repo"; sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+"/tests")
wm = importlib.import_module("rtl-wmbus_amd")
tmp = tempfile.mkdtemp(dir="/dev/shm")
caps = [wm.synth_capture(seed=9500 + i, n_samples=1 << 23, kinds=7, frames_per_s=20.0)[0] for i in range(8)]
for i, c in enumerate(caps): c.tofile(f"{tmp}/src{i}.cu8")
names=[]
for i in range(320):
    os.symlink(f"{tmp}/src{i % 8}.cu8", f"{tmp}/f{i:03d}.cu8"); names.append(f"f{i:03d}.cu8")
for rep in range(3):
    p = subprocess.run([wm.CLI_PATH, "-v", "-S"] + names, cwd=tmp, capture_output=True, env=dict(os.environ, WMBUS_FIXED_TS="1"))
    print(p.stderr.decode()[-900:])
import shutil; shutil.rmtree(tmp)
