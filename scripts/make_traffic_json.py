"""profiles/pmc_traffic.json from rocprofv3 FETCH_SIZE / WRITE_SIZE passes (scripts/pmc_passes.sh
<dir> fetch write): HBM-side bytes per launch of the dominant kernel.
Units and corrections per MI355X_MICROARCH.md §HBM: the counters are in KiB; on gfx950 FETCH_SIZE
reports half the bytes of a wide coalesced 16-B-per-lane stream and is uncalibrated for other shapes
(this kernel gathers 16-B pieces of 128-byte nodes and 48-byte triangles), so both the raw figure and
the doubled one (upper bound) are recorded; WRITE_SIZE is exact for 16-B stores."""
import csv, glob, json, sys, collections
root, kernel, out = sys.argv[1], sys.argv[2], sys.argv[3]
tot = collections.defaultdict(float)
cnt = collections.defaultdict(int)
for f in glob.glob(root + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if kernel in row["Kernel_Name"] and row["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU",
                                                                    "SQ_THREAD_CYCLES_VALU", "SQ_INSTS_VMEM_RD", "SQ_ACTIVE_INST_VALU",
                                                                    "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY"):
            tot[row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[row["Counter_Name"]] += 1
        if "k_traverse_packet" in row["Kernel_Name"] and row["Counter_Name"] in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM"):
            tot["cam_" + row["Counter_Name"]] += float(row["Counter_Value"])
            cnt["cam_" + row["Counter_Name"]] += 1
def per(name):
    return round(tot[name] / cnt[name]) if cnt[name] else None


fetch = tot["FETCH_SIZE"] * 1024 / max(1, cnt["FETCH_SIZE"])
write = tot["WRITE_SIZE"] * 1024 / max(1, cnt["WRITE_SIZE"])
# VALU pipe utilisation: SQ_ACTIVE_INST_VALU counts 4-cycle units (one per wave64 instruction on a 16-lane SIMD) summed
# over all SIMDs; SQ_BUSY_CYCLES counts cycles per shader engine (8 XCDs x 4), summed.  busy = 4 A / ((B / 32) * 1024 SIMDs)
active, busy = per("SQ_ACTIVE_INST_VALU"), per("SQ_BUSY_CYCLES")
valu_pipe_busy = round(active / (8.0 * busy), 4) if active and busy else None
wave_cycles, wait_inst = per("SQ_WAVE_CYCLES"), per("SQ_WAIT_INST_ANY")
json.dump({
    "kernel": kernel, "dispatches": cnt["FETCH_SIZE"],
    "source": "profiles/" + out.split("/")[-1] + " (rocprofv3 --pmc passes of scripts/" + ("pmc_c4.sh" if "c4" in out else "pmc_passes.sh") + ")",
    "valu_pipe_busy": valu_pipe_busy,
    "valu_pipe_busy_how": "SQ_ACTIVE_INST_VALU (4-cycle units over all SIMDs) x 4 / (SQ_BUSY_CYCLES / 32 shader engines x 1024 SIMDs)",
    "wave_cycles_waiting_frac": round(wait_inst / wave_cycles, 4) if wave_cycles and wait_inst else None,
    "valu_wave_insts_per_launch": per("SQ_INSTS_VALU"), "valu_thread_cycles_per_launch": per("SQ_THREAD_CYCLES_VALU"),
    "vmem_read_wave_insts_per_launch": per("SQ_INSTS_VMEM_RD"),
    "camera_ray_kernel": {"valu_wave_insts_per_launch": per("cam_SQ_INSTS_VALU"),
                          "salu_wave_insts_per_launch": per("cam_SQ_INSTS_SALU"),
                          "smem_wave_insts_per_launch": per("cam_SQ_INSTS_SMEM")},
    "fetch_bytes_per_launch_raw": round(fetch), "write_bytes_per_launch": round(write),
    "hbm_bytes_per_launch": round(2 * fetch + write),
    "note": "hbm_bytes_per_launch = 2 x FETCH_SIZE (gfx950 correction for 16-B-per-lane reads; an upper bound "
            "for this kernel's gathers) + WRITE_SIZE, averaged over the kernel's dispatches of the profiled "
            "program (scripts/prof_run.py: dragon stand-in 1080p, max_depth 5, 2 frames; scripts/bench_c4.py --prof: "
            "two 16.7 M-ray batches on the 10 M-triangle soup).  rocprofv3 sums a counter over the 8 XCDs.",
}, open(out, "w"), indent=1)
print(open(out).read())
