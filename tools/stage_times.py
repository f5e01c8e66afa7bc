#!/usr/bin/env python3
"""Dev tool: wall time (with device sync) of the stages of the eval forward at bench shapes."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_batch
from gvl_amd.config import make_opt
from gvl_amd.pdvc import build

from gvl_amd.tuning import enable_tuned_gemms
enable_tuned_gemms()
dev = torch.device("cuda:0")
opt = make_opt("anet_tsp_ssvg", num_queries=300, device="cuda")
torch.manual_seed(0)
model, criterion, _, _ = build(opt)
model = model.to(dev).eval()
for h in model.caption_head:
    h.graph_decode = True
dt = synth_batch(16, 100, 512, opt.vocab_size, 3, dev)


def timed(fn, n=10):
    for _ in range(3):
        r = fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, r


with torch.no_grad():
    t_enc, enc = timed(lambda: model.encode(dt))
    memory, tshapes, lsi, vr, mflat = enc
    qe = model.query_embed.weight
    pm = torch.ones(16, 300, dtype=torch.bool, device=dev)

    def dec():
        init_ref, tgt, ref, qpos = model.transformer.prepare_decoder_input_query(memory, qe)
        return model.transformer.forward_decoder(tgt, ref, memory, tshapes, lsi, vr, qpos, mflat, pm, False), init_ref
    t_dec, ((hs, inter), init_ref) = timed(dec)
    others = {'memory': memory, 'mask_flatten': mflat, 'spatial_shapes': tshapes, 'level_start_index': lsi,
              'valid_ratios': vr, 'proposals_mask': pm}
    t_cap, _ = timed(lambda: model.caption_head[-1].sample(hs[-1], inter[0], others))
    t_full_nocrit, _ = timed(lambda: model(dt, None, None, "queries", eval_mode=True))
    t_full, (out, loss) = timed(lambda: model(dt, criterion, None, "queries", eval_mode=True))
    t_crit, _ = timed(lambda: criterion(out, dt["video_target"]))
print(f"encode {t_enc:.2f} ms | decoder {t_dec:.2f} | captioner.sample {t_cap:.2f} | full w/o criterion {t_full_nocrit:.2f} | "
      f"criterion {t_crit:.2f} | full {t_full:.2f}")
if "--profile-criterion" in sys.argv:
    import cProfile, pstats
    pr = cProfile.Profile()
    with torch.no_grad():
        pr.enable()
        for _ in range(20):
            criterion(out, dt["video_target"])
        torch.cuda.synchronize()
        pr.disable()
    pstats.Stats(pr).sort_stats("cumtime").print_stats(28)
