import sys, time, numpy as np
sys.path.insert(0, '.')
import bench, llcomp_amd as mi
frames = bench.make_frames("g3", 32, 0)
F,h,w,c = frames.shape
pinned = mi.PinnedBuffer(frames.size); pinned.array[:] = frames.reshape(-1)
views = [pinned.array[i*h*w*c:(i+1)*h*w*c].reshape(h,w,c) for i in range(F)]
for depth, inflight, vt, verify in ((8,3,4,True),(8,3,4,False),(12,4,4,True),(12,5,6,True),(16,6,6,True),(16,6,6,False),(6,2,4,True)):
    st = mi.Stream(w,h,c,480,1,True,depth=depth)
    jobs = views+views
    lens, done, busy = mi.pipeline_roundtrip(st, jobs, max_encodes_in_flight=inflight, verify=verify, verify_threads=vt)
    st.close()
    n=len(jobs)
    print(f"depth {depth} inflight {inflight} threads {vt} verify {verify}: {(n-4)*w*h/1e6/(done[-1]-done[3]):.0f} MPix/s busy {busy}", flush=True)
