"""GB/s of the HBM-bound kernels of BASELINE config 5 (G forward, 4 x 3 x 512 x 512 -> 4 x 3 x 2048 x 2048) from the condensed
kernel trace: algorithmic bytes (read the input once, write the output once, weights negligible) / mean kernel time."""
import csv, sys
src, out = sys.argv[1], sys.argv[2]
B = 4
algo = {   # kernel-name substring -> (layer, algorithmic bytes per launch)
    "conv_rgb_in_kernel": ("embed 3->256 @512x512 (reads 3 ch, writes 256 ch)", B * 512 * 512 * (3 + 256) * 4),
    "conv_rgb_out_kernel": ("upsample.4 256->3 @2048x2048 (reads 256 ch, writes 3 ch)", B * 2048 * 2048 * (256 + 3) * 4),
    "meanshift_fwd_kernel": None,
}
rows = list(csv.DictReader(open(src)))
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "layer", "launches_per_forward", "avg_us", "algorithmic_MB", "GB_per_s", "frac_of_6.29_TB_s"])
    for r in rows:
        name = r["kernel"]
        for key, val in algo.items():
            if key in name:
                n = int(r[[c for c in r if c.startswith("launches")][0]]) // 2
                us = float(r["avg_us"])
                if val is None:   # MeanShift: two launches per forward with different sizes (sub_mean @512^2, add_mean @2048^2)
                    grid = int(r["grid_threads"])
                    px = B * 512 * 512 if grid < 2000000 and us < 50 else B * 2048 * 2048
                    layer, nbytes = (f"MeanShift 3->3 ({'sub_mean @512x512' if px == B * 512 * 512 else 'add_mean @2048x2048'})", px * 6 * 4)
                else:
                    layer, nbytes = val
                gbs = nbytes / us / 1e3
                w.writerow([name[:60], layer, n, us, round(nbytes / 1e6, 1), round(gbs, 1), round(gbs / 6290, 3)])
