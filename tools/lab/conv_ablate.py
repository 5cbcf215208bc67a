import os, sys, subprocess
for d in (15, 15 + 16, 15 + 32, 15 + 64, 15 + 128, 15 + 16 + 32 + 64, 255):
    env = dict(os.environ, RP_CONV_DBG=str(d), QUICK="1")
    out = subprocess.run([sys.executable, "tools/conv3x3_time.py"], env=env, capture_output=True, text=True).stdout
    print("dbg=%2d" % d, [l for l in out.split("\n") if l.startswith("own plain") or l.startswith("own + BN/ReLU on load  ")])
