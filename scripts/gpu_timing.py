"""Debug: per-phase cycle breakdown of the lock-step path kernel (library built with -DHPT_TIMING, whose
instrumented kernel accumulates clock64() deltas of lane 0 of every wave into the work counters)."""
import importlib, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench
abi = importlib.import_module("pbrt-v2_amd.abi")
hpt = importlib.import_module("pbrt-v2_amd.hpt")
for w in sys.argv[1:]:
    scene, desc = bench.load_workload(w, 16)
    rd = abi.copy_struct(scene.render)
    rd.count_work = 1
    dev = hpt.DeviceScene(scene, 0)
    _, st = dev.render(scene.camera, rd)
    _, st = dev.render(scene.camera, rd)
    tot = st.closest_rays + st.bad_samples + st.shadow_rays + st.camera_samples
    print(w, "kernel_ms %.1f" % st.kernel_ms, "trace_ext %.3f shade_ext %.3f trace_sh_mis %.3f shade_sh_mis %.3f" % (
        st.closest_rays / tot, st.bad_samples / tot, st.shadow_rays / tot, st.camera_samples / tot),
        "cycles/wave-total %.3g" % tot, "| of shade_ext: geom+NEE %.3f  whole hit block %.3f" % (st.nodes_visited / max(st.bad_samples, 1), st.tris_tested / max(st.bad_samples, 1)))
