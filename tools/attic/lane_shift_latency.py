import os, sys, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
import llcomp_amd as mi
fr4k = bench.make_frames("g3", 1, 0)
c2 = bench.make_frames("g3", 1, 0, w=1920, h=1080, c=3)
for name, fr, tw, th, planar, shifts in (("4k_480x1_planar", fr4k, 480, 1, True, (6,5,4,3,2)), ("c2_1920x1_interleaved", c2, 1920, 1, False, (4,3,2,1,0)), ("c2_1920x1_planar", c2, 1920, 1, True, (5,4,3,2,1,0)),
                                         ("4k_64x64_planar_1frame", fr4k, 64, 64, True, (6,5,4,3,2))):
    for sh in shifts:
        os.environ["LLCOMP_MI_LANE_SHIFT"] = str(sh); mi.reload_tuning()
        m = bench.measure(fr, tw, th, planar, 1, 12, 2, 0)
        print(json.dumps({"case": name, "lane_shift": sh, "ms": round(m["dt"]/m["steps"]*1e3,3), "mpix": round(m["mpix"],1), "enc": round(m["prof"]["k_encode_slices"]/m["steps"],3), "dec": round(m["prof"]["k_decode_slices"]/m["steps"],3)}), flush=True)
