"""How much of a whole wave's life is its ragged end?  A build with -DRTM_EXP_TRIPS=1 (profiles/build_ab.sh trips "-DRTM_EXP_TRIPS=1")
reports 64 x (trips of the wave) in rtm_stats.draws; casts counts the live lanes of every trip, so 1 - casts / draws is the share of
lane-trips spent by lanes that had finished their pixel (or lie outside the image) while their wave was still running.
    RTM_LIB_OVERRIDE=ab_libs/librtm_trips.so python profiles/exp/ragged_end.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import raytracingmin_amd as rtm
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
data = rtm.LoadData(os.path.join(root, "scenes", "cornellBoxSetting.json")).data
for (w, h, s, ss, tag) in ((1920, 1080, 64, 4, "headline 1080p x 1024 spp"), (512, 512, 16, 4, "512 x 512 x 256 spp"), (1920, 1080, 4, 4, "1080p x 64 spp")):
    data.width, data.height, data.samples, data.superSamples = w, h, s, ss
    for variant, name in ((2, "whole tiles only (variant 2)"), (9, "every tile split (variant 9)"),
                          (0, "default (stealing + tail split)")):
        r = rtm.Renderer(data, mode="repaired", max_bounces=8, seed=0x5EED, variant=variant)
        _, st = r.render_rows_device(want=("f32",), stats=True)
        print(f"{tag:28s} {name:30s} lane-trips {st['draws']:>14d}  live {st['casts']:>14d}  idle share {1 - st['casts'] / st['draws']:.4f}  kernel {st['kernel_ms']:.2f} ms")
