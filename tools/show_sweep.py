import json, sys
for line in open(sys.argv[1]):
    line = line.strip()
    if not line.startswith('{'):
        if 'amdgpu.ids' not in line and line: print(line[:200])
        continue
    r = json.loads(line); c = r['config']; k = r['kernel_ms_per_step']
    print(f"{c['content']:4s} F={c['frames_per_step_per_gpu']:2d} tile={c['tile_w']}x{c['tile_h']}{'p' if c['planar'] else 'i'} slices/f={c['slices_per_frame']:6d} value={r['value']:9.1f} MPix/s ms/step={r['ms_per_step']:8.2f} ratio={c['compression_ratio']:.4f} enc={k['k_encode_slices']:.3f} dec={k['k_decode_slices']:.3f} model={k['k_model_fwd']:.3f} inv={k['k_model_inv']:.3f} clr={k['clear_states_enc']:.3f} pack={k['scan+pack']:.3f} frac={r['roofline']['frac']}")
