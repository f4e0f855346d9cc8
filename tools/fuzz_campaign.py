#!/usr/bin/env python3
"""One-off differential fuzz campaign on the GPU box: tests/test_gpu_parity.py::test_fuzz_random_scenes with N seeds (scene
`seed` is numpy default_rng(1000 + seed)) and ::test_fuzz_extension_modes with N // 3 seeds (default_rng(7000 + seed), per-lane and
wavefront form), every case against the CPU oracle on all parity planes.  Writes the seeds and the result as JSON so that a
campaign is a reproducible artifact (profiles/rNN_experiments/fuzz_campaign.json), not a line in a log.
   python tools/fuzz_campaign.py <n_seeds> <out.json> [first_seed]"""
import importlib, json, os, subprocess, sys, time
import xml.etree.ElementTree as ET
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
n, out = int(sys.argv[1]), sys.argv[2]
xml = out + ".junit.xml"
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
env = dict(os.environ, RT_FUZZ_SEEDS=str(n), RT_FUZZ_FIRST=str(first))
t0 = time.time()
# (a line a minute while the cases run: the GPU box's watchdog takes seven silent minutes for a hang)
proc = subprocess.Popen([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-q", "-m", "gpu", "-k", "fuzz",
                         "--junitxml", xml, "-p", "no:cacheprovider"], env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
import threading
chunks = []
reader = threading.Thread(target=lambda: chunks.extend(iter(lambda: proc.stdout.read(4096), "")), daemon=True)
reader.start()
while proc.poll() is None:
    try:
        proc.wait(timeout=60)
    except subprocess.TimeoutExpired:
        print("fuzz campaign: %d s, %d bytes of pytest output so far" % (time.time() - t0, sum(len(c) for c in chunks)), flush=True)
reader.join(timeout=10)


class _R:
    returncode, stdout = proc.returncode, "".join(chunks)


r = _R
cases = list(ET.parse(xml).getroot().iter("testcase"))
bad = [c.get("name") for c in cases if any(ch.tag in ("failure", "error") for ch in c)]
skipped = [c.get("name") for c in cases if any(ch.tag == "skipped" for ch in c)]
res = {"campaign": "differential fuzz, HIP path vs CPU oracle, all parity planes", "date": time.strftime("%Y-%m-%d %H:%M:%S UTC", time.gmtime()),
       "kernel_code_hash": importlib.import_module("cuda-raytracing_amd").library_hash(),
       "test_fuzz_random_scenes": {"seeds": [first, first + n], "rng": "numpy.random.default_rng(1000 + seed)", "gpu_built_tree": "seed % 3 == 2"},
       "test_fuzz_extension_modes": {"seeds": [first, first + max(4, n // 3)], "rng": "numpy.random.default_rng(7000 + seed)", "forms": ["per-lane", "two-launch (RT_EX_SPLIT=1)", "wavefront"]},
       "cases": len(cases), "passed": len(cases) - len(bad) - len(skipped), "failed": bad, "skipped": skipped,
       "pytest_exit_code": r.returncode, "pytest_summary": r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "", "seconds": round(time.time() - t0, 1)}
json.dump(res, open(out, "w"), indent=1)
os.remove(xml)
print(json.dumps({k: res[k] for k in ("cases", "passed", "failed", "pytest_summary", "seconds")}))
sys.exit(0 if r.returncode == 0 and not bad else 1)
