#!/usr/bin/env python3
"""One-off campaign of the ADVERSARIAL differential fuzz on the GPU box: tests/test_gpu_parity.py::test_fuzz_adversarial_scenes with N
seeds (numpy default_rng(31000 + seed): awkward mesh kinds x instance forms x cameras, every parity plane against the CPU oracle) and
::test_fuzz_adversarial_extension_modes (default_rng(47000 + seed)), ::test_fuzz_adversarial_refit_and_rebuild (default_rng(59000 + seed))
and ::test_fuzz_adversarial_api_sequences (default_rng(67000 + seed)) with N // 3 seeds each.  Writes seeds and result as JSON
(profiles/rNN_experiments/fuzz_adversarial_*.json).
   python tools/fuzz_adversarial_campaign.py <n_seeds> <out.json> [first_seed]"""
import importlib, json, os, re, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
n, out = int(sys.argv[1]), sys.argv[2]
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
env = dict(os.environ, RT_FUZZ_ADV_SEEDS=str(n), RT_FUZZ_ADV_FIRST=str(first))
t0 = time.time()
proc = subprocess.Popen([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-q", "-m", "gpu", "-k", "adversarial",
                         "-rf", "--maxfail", "50", "-p", "no:cacheprovider"], env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
chunks = []
reader = threading.Thread(target=lambda: chunks.extend(iter(lambda: proc.stdout.read(4096), "")), daemon=True)
reader.start()
while proc.poll() is None:
    try:
        proc.wait(timeout=60)
    except subprocess.TimeoutExpired:       # (a line a minute: the GPU box's watchdog takes seven silent minutes for a hang)
        print("adversarial fuzz campaign: %d s, %d bytes of pytest output so far" % (time.time() - t0, sum(len(c) for c in chunks)), flush=True)
reader.join(timeout=10)
text = "".join(chunks)
summary = text.strip().splitlines()[-1] if text.strip() else ""
failed = re.findall(r"^FAILED (\S+)", text, re.M)
m = re.search(r"(\d+) passed", summary)
passed = int(m.group(1)) if m else 0
res = {"campaign": "adversarial differential fuzz, HIP path vs CPU oracle: awkward mesh kinds (lattice, zero-area, piles of coincident triangles, 1e18 / 1e-20 "
                   "coordinates, slivers, non-finite vertices) x instance forms (identity, translated, lattice steps, signed zeros, quarter turns, any; unit, "
                   "power-of-two, mirrored, 1e2..1e4, 1e-4..1e-2 scales) x cameras (any, axis-parallel central rays, on the lattice, inside)",
       "date": time.strftime("%Y-%m-%d %H:%M:%S UTC", time.gmtime()),
       "kernel_code_hash": importlib.import_module("cuda-raytracing_amd").library_hash(),
       "test_fuzz_adversarial_scenes": {"seeds": [first, first + n], "rng": "numpy.random.default_rng(31000 + seed)", "gpu_built_tree": "seed % 3 == 2",
                                        "checked": "RGB + hit ids of the production kernel, six planes of the instrumented kernel, a batch of four frames through view records"},
       "test_fuzz_adversarial_extension_modes": {"seeds": [first, first + max(4, n // 3)], "rng": "numpy.random.default_rng(47000 + seed)", "checked": "RGB + total pops; the default form, then the two-launch form (even seeds) or the wavefront form (odd seeds)"},
       "test_fuzz_adversarial_refit_and_rebuild": {"seeds": [first, first + max(4, n // 3)], "rng": "numpy.random.default_rng(59000 + seed)",
                                                   "checked": "all planes as uploaded, after a refit to another awkward mesh, after a device rebuild from a third"},
       "test_fuzz_adversarial_api_sequences": {"seeds": [first, first + max(4, n // 3)], "rng": "numpy.random.default_rng(67000 + seed)",
                                               "checked": "six random scene-changing calls (refit / rebuild incl. growth / instance update / upload again), after each a frame "
                                                          "through a random entry point (planes, hit ids, batch of four with two poses, stripes + rt_unstripe, two default-stream frames)"},
       "cases": passed + len(failed), "passed": passed, "failed": failed, "pytest_exit_code": proc.returncode, "pytest_summary": summary,
       "seconds": round(time.time() - t0, 1)}
json.dump(res, open(out, "w"), indent=1)
if failed or proc.returncode:
    print(text[-6000:])
print(json.dumps({k: res[k] for k in ("cases", "passed", "failed", "pytest_summary", "seconds")}))
sys.exit(0 if proc.returncode == 0 and not failed else 1)
