"""One C2 launch under rocprofv3 --kernel-trace: prints the kernel timeline (start / end in ms relative to the first
kernel of the traced launch) so that the overlap of the two lanes can be read off.
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python3 tools/lane_trace.py run [lanes] [case]
    python tools/lane_trace.py show gpurun_out/trace"""
import os, sys, glob, csv
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == "run":
    import light_transport_amd as lt
    from tests import scenes as S
    lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    case = sys.argv[3] if len(sys.argv) > 3 else "c2"
    if case in ("teapot", "cow", "pumpkin"):      # the reference's OBJ assets (fixture G10) in a closed box
        import numpy as np
        g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "g10_obj_meshes.npz"))
        prob, n, mode = S.obj_in_box(g[case + "_verts"], g[case + "_faces"])[0], 10 ** 7, "log"
    else:
        prob, n, mode = {"c2": lambda: (S.slab(n=256, voxel=0.1), 10 ** 7, "log"), "c3": lambda: (S.two_layer(n=256, voxel=0.05), 10 ** 7, "log"),
                         "c4": lambda: (S.cornell(256), 10 ** 7, "auto"), "c5": lambda: (S.two_layer(n=512, voxel=0.025), 12500000, "log"),
                         "sphere": lambda: (S.sphere_in_box(4, split_method=0)[0], 10 ** 7, "log")}[case]()
    ctx = lt.Context(0)
    prob.apply(ctx, "f64"); ctx.set_tally_mode(mode); ctx.set_overlap(lanes)
    for r in range(3):
        ctx.zero_tally(); ctx.launch(n, seed=r); ctx.sync()
        print("launch %d: %.2f ms" % (r, ctx.last_kernel_ms()), flush=True)
    ctx.close()
else:
    f = sorted(glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    walks = [i for i, r in enumerate(rows) if "walk_kernel" in r["Kernel_Name"]]
    # last launch = after the last gap > 20 ms between walk kernels ... simply take kernels after the last k_grid/zero fill
    starts = [int(r["Start_Timestamp"]) for r in rows]
    # find the start of the last launch: the last walk whose predecessor walk ended > 1 ms earlier than it started and
    # which follows a reduce of a previous launch; simpler: split on gaps > 3 ms of idle
    ends = [int(r["End_Timestamp"]) for r in rows]
    cut = 0
    busy_end = ends[0]
    for i in range(1, len(rows)):
        if starts[i] - busy_end > 3e6:
            cut = i
        busy_end = max(busy_end, ends[i])
    t0 = starts[cut]
    for r in rows[cut:]:
        name = r["Kernel_Name"]
        short = "walk" if "walk_kernel" in name else name[name.index("k_log_"):].split("(")[0].split("<")[0] if "k_log_" in name else name[:30]
        print("%-16s q%-3s %8.3f -> %8.3f  (%.3f ms)" % (short, r.get("Queue_Id", "?"), (int(r["Start_Timestamp"]) - t0) / 1e6,
                                                        (int(r["End_Timestamp"]) - t0) / 1e6,
                                                        (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
