"""CPU restatement of the low-to-high sample-count reconstruction stage (rt_render_guides,
rt_reconstruct of include/rt.h).  TEST INFRASTRUCTURE ONLY: imported by tests/, by
__graft_entry__.smoke() and by bench.py's checks, never by raytrace_amd/.

PARITY UNPINNED: the reference names such a stage as ongoing work (README.md:7, "Super Resolution
GAN ... similar to DLSS"; the frame pairs it would train on are laid out in cuda/main.cu:557-576)
and holds no code, weights or output of it, so there is nothing of the reference's to be identical
to.  What is checked instead: the kernels against this restatement (same arithmetic in double), the
guides against the closest-hit checker, and the result against a high-sample-count frame as PSNR.

The filter is the edge-avoiding a-trous wavelet transform of Dammertz, Sewtz, Hanika, Lensch
(HPG 2010): N passes of a 5x5 B3-spline kernel whose taps stand 2^i pixels apart in pass i, every tap
weighted by how well it agrees with the centre pixel in first-hit normal, first-hit distance,
first-hit object and (demodulated) colour, the colour tolerance halving from pass to pass.
"""
import numpy as np

B3 = np.array([1 / 16, 1 / 4, 3 / 8, 1 / 4, 1 / 16])
DEFAULTS = dict(iterations=4, sigma_color=0.0, sigma_normal=16.0, sigma_depth=0.05, albedo_floor=0.02)   # rt_reconstruct's


def centre_rays(cam, W, H):
    """The ray through the centre of every pixel from the lens centre (camera.h:40-45 with a zero
    lens offset; s = (i + 0.5) / (W - 1) as main.cc:117-118 with the jitter at its mean), as float32
    values: rows bottom-up like the framebuffer.  cam: the 22 doubles of Oracle.camera."""
    o, llc, hor, ver = cam[0:3], cam[3:6], cam[6:9], cam[9:12]
    f32 = lambda a: np.asarray(a, np.float32)
    o32, rel, hor, ver = f32(o), f32(llc - o), f32(hor), f32(ver)       # the device camera: llc - origin rounded once
    inv_w, inv_h = np.float32(1.0 / (W - 1)), np.float32(1.0 / (H - 1))
    s = (np.arange(W, dtype=np.float32) + np.float32(0.5)) * inv_w
    t = (np.arange(H, dtype=np.float32) + np.float32(0.5)) * inv_h
    # fused multiply-adds on the device: evaluated in double and rounded per step here
    step1 = (rel[None, None, :].astype(np.float64) + s[None, :, None].astype(np.float64) * hor[None, None, :].astype(np.float64)).astype(np.float32)
    d = (step1.astype(np.float64) + t[:, None, None].astype(np.float64) * ver[None, None, :].astype(np.float64)).astype(np.float32)
    rays = np.concatenate([np.broadcast_to(o32, d.shape), d], axis=2)
    return rays.reshape(-1, 6).astype(np.float64)


def guides(oracle, flat, cam, W, H, t_min=0.001):
    """(g0, g1, ids): g0 = first-hit normal and distance, g1 = first-hit albedo, ids = list position or -1."""
    hit = oracle.trace(flat, centre_rays(cam, W, H), t_min=t_min)
    n = W * H
    g0, g1 = np.zeros((n, 4)), np.ones((n, 3))
    for k in np.where(hit["id"] >= 0)[0]:
        m = int(flat.prim_mat[hit["id"][k]])
        g0[k, :3], g0[k, 3] = hit["n"][k], hit["t"][k]
        if flat.mat_type[m] == 2:                                  # a dielectric has no colour
            g1[k] = 1.0
        elif flat.mat_tex[m] >= 0 and flat.mat_type[m] != 1:        # (metal reads no texture)
            g1[k] = oracle.texture_value(flat, int(flat.mat_tex[m]), hit["uv"][k, 0], hit["uv"][k, 1], hit["p"][k])
        else:
            g1[k] = flat.mat_param[m, :3]
    return g0.reshape(H, W, 4), np.maximum(g1, 0.0).reshape(H, W, 3), hit["id"].reshape(H, W)


def reconstruct(rgba, g0, g1, ids, iterations=4, sigma_color=0.0, sigma_normal=16.0, sigma_depth=0.05, albedo_floor=0.02):
    """rgba: [H, W, 4] sums of sample radiance + sample count.  Returns [H, W, 4] in the same
    convention (rgb = reconstructed mean x sample count).  sigma_color 0: 1.6 / sqrt(samples of the pixel)."""
    H, W = ids.shape
    spp = rgba[..., 3:4].astype(np.float64)
    alb = np.maximum(g1.astype(np.float64), albedo_floor)
    c = rgba[..., :3].astype(np.float64) / np.maximum(spp, 1.0) / alb
    nrm, dist, hit = g0[..., :3].astype(np.float64), g0[..., 3].astype(np.float64), ids >= 0
    for it in range(iterations):
        step = 1 << it
        sc = sigma_color if sigma_color > 0 else 1.6 / np.sqrt(np.maximum(spp[..., 0], 1.0))
        inv_sc2 = 1.0 / (sc * 0.5 ** it) ** 2
        acc, wsum = np.zeros_like(c), np.zeros((H, W))
        for dy in range(-2, 3):
            for dx in range(-2, 3):
                h = B3[dy + 2] * B3[dx + 2]
                ys, xs = np.arange(H) + dy * step, np.arange(W) + dx * step
                inside = ((ys >= 0) & (ys < H))[:, None] & ((xs >= 0) & (xs < W))[None, :]
                yc, xc = np.clip(ys, 0, H - 1), np.clip(xs, 0, W - 1)
                qc, qn, qd, qid = c[yc][:, xc], nrm[yc][:, xc], dist[yc][:, xc], ids[yc][:, xc]
                w = np.where(inside & (qid == ids), h, 0.0)
                cosn = np.clip(np.sum(nrm * qn, axis=2), 1e-6, 1.0)
                geo = np.exp2(sigma_normal * np.log2(cosn)) * np.exp(-np.abs(dist - qd) / (sigma_depth * np.maximum(dist, 1e-6)))
                w = w * np.where(hit, geo, 1.0)
                w = w * np.exp(-np.sum((c - qc) ** 2, axis=2) * inv_sc2)
                acc += w[..., None] * qc
                wsum += w
        c = acc / wsum[..., None]                                   # the centre tap always counts: wsum >= 9/64
    out = np.concatenate([c * alb * spp, spp], axis=2)
    return out


def upscale(rgba_low, g0l, g1l, idl, g0h, g1h, idh, sigma_normal=16.0, sigma_depth=0.05, albedo_floor=0.02):
    """Joint bilateral upsampling (Kopf et al. 2007) of a low-resolution frame onto the grid of the high-resolution
    guides, as rt_upscale does it: 3x3 low-resolution taps around the pixel's own position, tent weights of width 1.5
    low pixels times guide agreement, on demodulated colours; remodulated by the high-resolution albedo; tent weights
    alone where no tap agrees.  Returns [H, W, 4]."""
    lh, lw = idl.shape
    hh, hw = idh.shape
    spp_l = rgba_low[..., 3].astype(np.float64)
    c = rgba_low[..., :3].astype(np.float64) / np.maximum(spp_l, 1.0)[..., None] / np.maximum(g1l.astype(np.float64), albedo_floor)
    f32 = np.float32
    ux = ((np.arange(hw, dtype=f32) + f32(0.5)) * (f32(lw - 1) / f32(hw - 1)) - f32(0.5)).astype(np.float64)
    uy = ((np.arange(hh, dtype=f32) + f32(0.5)) * (f32(lh - 1) / f32(hh - 1)) - f32(0.5)).astype(np.float64)
    bx, by = np.floor(ux + 0.5).astype(int), np.floor(uy + 0.5).astype(int)
    nrm, dist, hit = g0h[..., :3].astype(np.float64), g0h[..., 3].astype(np.float64), idh >= 0
    acc, wsum, bacc, bsum = np.zeros((hh, hw, 3)), np.zeros((hh, hw)), np.zeros((hh, hw, 3)), np.zeros((hh, hw))
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            qx, qy = np.clip(bx + dx, 0, lw - 1), np.clip(by + dy, 0, lh - 1)
            tent = np.maximum(1.0 - np.abs(by + dy - uy) / 1.5, 0.0)[:, None] * np.maximum(1.0 - np.abs(bx + dx - ux) / 1.5, 0.0)[None, :]
            qc, qn, qd, qid = c[qy][:, qx], g0l[qy][:, qx][..., :3].astype(np.float64), g0l[qy][:, qx][..., 3].astype(np.float64), idl[qy][:, qx]
            bacc += tent[..., None] * qc
            bsum += tent
            cosn = np.clip(np.sum(nrm * qn, axis=2), 1e-6, 1.0)
            geo = np.exp2(sigma_normal * np.log2(cosn)) * np.exp(-np.abs(dist - qd) / (sigma_depth * np.maximum(dist, 1e-6)))
            w = np.where(qid == idh, tent, 0.0) * np.where(hit, geo, 1.0)
            acc += w[..., None] * qc
            wsum += w
    agreed = wsum > 1e-4
    res = np.where(agreed[..., None], acc / np.where(agreed, wsum, 1.0)[..., None], bacc / bsum[..., None])
    spp = spp_l[np.clip(by, 0, lh - 1)][:, np.clip(bx, 0, lw - 1)]
    return np.concatenate([res * np.maximum(g1h.astype(np.float64), albedo_floor) * spp[..., None], spp[..., None]], axis=2)


def bilinear(rgba_low, hw, hh):
    """Plain tent-weighted upsampling of the mean radiance with the same pixel mapping (the baseline rt_upscale is held against)."""
    lh, lw = rgba_low.shape[:2]
    m = rgba_low[..., :3].astype(np.float64) / np.maximum(rgba_low[..., 3:4], 1.0)
    ux = (np.arange(hw) + 0.5) * ((lw - 1) / (hw - 1)) - 0.5
    uy = (np.arange(hh) + 0.5) * ((lh - 1) / (hh - 1)) - 0.5
    x0, y0 = np.clip(np.floor(ux).astype(int), 0, lw - 2), np.clip(np.floor(uy).astype(int), 0, lh - 2)
    fx, fy = np.clip(ux - x0, 0, 1)[None, :, None], np.clip(uy - y0, 0, 1)[:, None, None]
    a, b, c2, d = m[y0][:, x0], m[y0][:, x0 + 1], m[y0 + 1][:, x0], m[y0 + 1][:, x0 + 1]
    out = (a * (1 - fx) + b * fx) * (1 - fy) + (c2 * (1 - fx) + d * fx) * fy
    spp = float(rgba_low[0, 0, 3])
    return np.concatenate([out * spp, np.full((hh, hw, 1), spp)], axis=2)


def psnr(img, ref, spp_img, spp_ref):
    """Peak signal-to-noise ratio of the displayed images: mean radiance through the reference's gamma 2 and
    clamp (color.h:17-23), peak 1."""
    a = np.sqrt(np.clip(img[..., :3] / spp_img, 0.0, 1.0))
    b = np.sqrt(np.clip(ref[..., :3] / spp_ref, 0.0, 1.0))
    return float(10.0 * np.log10(1.0 / np.mean((a - b) ** 2)))
