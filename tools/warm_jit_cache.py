"""Loads every fixture and every seeded feature-rich test model once on this machine (no GPU needed): models whose layer
shapes are outside nam_wn_reg_kernel's ahead-of-time tables get the kernel compiled for them here, and the code objects
land in neuralampmodelercore_amd/lib/jit/ — next to the library, so they travel with it (the GPU box then finds them by
hash instead of compiling). Run by __graft_entry__.build(). --prune: delete cache files no load of this run asked for
(code objects of older kernel sources).

Deployment: `python tools/warm_jit_cache.py --only path/to/a.nam path/to/b.nam ...` on a machine WITH hipcc (no GPU needed) builds the
code objects of exactly those models (both tanh modes) into lib/jit/ (or $NAM_HIP_JIT_CACHE); ship that directory with the library and
the serving hosts need neither a compiler nor the kernel sources ("Kernel selection")."""
import glob, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import neuralampmodelercore_amd as nam
import make_synthetic_models as msm


def _load(job):
    p, ft = job
    m = nam.get_dsp(p, fast_tanh=ft)
    return bool(m.info.has_a1_kernel & 16)


def main():
    t0 = time.time()
    n = acc = 0
    cache = os.environ.get("NAM_HIP_JIT_CACHE") or os.path.join(ROOT, "neuralampmodelercore_amd", "lib", "jit")
    if "--only" in sys.argv:  # deployment: exactly the models named on the command line
        paths = [a for a in sys.argv[sys.argv.index("--only") + 1:] if not a.startswith("--")]
        for p in paths:
            for ft in (False, True):
                m = nam.get_dsp(p, fast_tanh=ft)
                failed = bool(m.info.has_a1_kernel & 32)
                print(f"{p} fast_tanh={ft}: nam_wn_reg_kernel {'compiled for its shapes / built in' if (m.info.has_a1_kernel & 16) and not failed else 'not used' if not failed else 'COMPILE FAILED (see stderr)'}")
        print(f"warm_jit_cache: {len(paths)} models, {len(glob.glob(os.path.join(cache, '*.hsaco')))} code objects in {cache}")
        return
    stamp = time.time() - 1.0  # code objects a load below neither built nor found are stale (their sources changed): pruned
    # (round 6: every nam_wn_reg_kernel model gets a code object of its own — its programs compiled in —, ~6 s each: the
    # loads run in worker processes, one per core; the cache publishes by rename, concurrent builds of one key are benign)
    import multiprocessing as mp
    jobs = []
    for p in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "models", "*.nam"))):
        for ft in (False, True):
            jobs.append((p, ft))
    with tempfile.TemporaryDirectory() as tmp:
        for seed in range(50):  # tests/test_gpu_breadth.py: FEATURED_SEEDS (40 ..: with a post-stack head)
            p = os.path.join(tmp, f"featured_{seed}.nam")
            msm.write_featured(p, 7000 + seed, wr_shapes=bool(seed % 2), post_head=seed >= 40)
            jobs.append((p, seed % 3 == 0))
        workers = max(1, min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 8))
        with mp.get_context("spawn").Pool(workers) as pool:
            res = pool.map(_load, jobs)
    n = len(jobs)
    acc = sum(1 for (p, _), ok in zip(jobs, res) if ok and os.path.basename(p).startswith("featured_"))
    pruned = 0
    for f in glob.glob(os.path.join(cache, "*")) if "--prune" in sys.argv else []:  # (a hit refreshes the file's time stamp, wr_jit.cpp)
        if os.path.getmtime(f) < stamp:
            os.remove(f)
            pruned += 1
    print(f"warm_jit_cache: {n} loads in {time.time() - t0:.1f} s, {acc} / 50 feature-rich models on nam_wn_reg_kernel, "
          f"{len(glob.glob(os.path.join(cache, '*.hsaco')))} code objects in {cache} ({pruned} stale files pruned)")


if __name__ == "__main__":
    main()
