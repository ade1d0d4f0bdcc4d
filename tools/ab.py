"""A/B two library builds on the config-2 batch in separate processes (dev tool)."""
import subprocess, sys
for lib in sys.argv[1:]:
    out = subprocess.run([sys.executable, "tools/quick_bench.py", "1024"], env={**__import__("os").environ, "MTRPL_LIB_PATH": lib}, capture_output=True, text=True)
    print(lib.split("/")[-1], "|", out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:])
