"""bench.py's workloads and the objects its timed step is made of (round 6: out of bench.py, which is the headline driver).

  WORKLOADS      BASELINE.json's configs as (width, height, sampling, mode, colour transform, images per GPU[, dct_scale])
  build_variants the synthetic coefficients of a workload (tests/synth.py) and their batch descriptors
  Shard          one rank's images as launch groups (jpgpu batches) over one pair of device arenas: decode() is the timed step
  PixelGather    north_star's final gather of the pixels to rank 0 (RCCL point-to-point per launch group)"""
import numpy as np

WORKLOADS = {
    # name: (width, height, sampling, mode, colour transform, default images per GPU at N = 1[, dct_scale])
    "1080p-420": (1920, 1080, [(2, 2), (1, 1), (1, 1)], "ycbcr", "YCbCr", 256),
    "2160p-420": (3840, 2160, [(2, 2), (1, 1), (1, 1)], "ycbcr", "YCbCr", 64),
    "1080p-444": (1920, 1080, [(1, 1), (1, 1), (1, 1)], "ycbcr", "YCbCr", 256),
    "1080p-422": (1920, 1080, [(2, 1), (1, 1), (1, 1)], "ycbcr", "YCbCr", 256),
    "1080p-440": (1920, 1080, [(1, 2), (1, 1), (1, 1)], "ycbcr", "YCbCr", 256),
    "1080p-411": (1920, 1080, [(4, 1), (1, 1), (1, 1)], "ycbcr", "YCbCr", 256),  # UpsamplerGeneric layout (fusedgen)
    "1080p-gray": (1920, 1080, [(1, 1)], "gray", "Grayscale", 256),
    "1080p-cmyk": (1920, 1080, [(1, 1)] * 4, "cmyk", "CMYK", 192),
    # layouts the reference's own fixtures have (jpg-cmyk-2.jpg; YCCK as Photoshop writes it) and reduced-size decodes (Decoder::scale)
    "1080p-cmyk-2211": (1920, 1080, [(2, 2), (1, 1), (1, 1), (1, 1)], "cmyk", "CMYK", 192),
    "1080p-ycck-2212": (1920, 1080, [(2, 2), (1, 1), (1, 1), (2, 2)], "ycck", "YCCK", 192),
    "1080p-420-scale4": (1920, 1080, [(2, 2), (1, 1), (1, 1)], "ycbcr", "YCbCr", 256, 4),
    "1080p-420-scale2": (1920, 1080, [(2, 2), (1, 1), (1, 1)], "ycbcr", "YCbCr", 256, 2),
    "1080p-420-scale1": (1920, 1080, [(2, 2), (1, 1), (1, 1)], "ycbcr", "YCbCr", 256, 1),
    "1080p-444-scale4": (1920, 1080, [(1, 1), (1, 1), (1, 1)], "ycbcr", "YCbCr", 256, 4),
    "1080p-444+gray": (1920, 1080, None, "mixed", None, 1024),  # BASELINE configs[4]: batch 1024 (512 + 512), two fused launch groups
}
CONFIG3_WORKLOAD, CONFIG3_IMAGES_TOTAL = "2160p-420", 4096
E2E_SHARDED_TOTAL = 4096  # files of the e2e leg at N > 1 (north_star's batch), sharded over the ranks


class Shard:
    """One rank's images as `n_sub` launch groups (jpgpu batches) over one pair of device arenas."""

    def __init__(self, J, torch, variants, n_img, n_sub, device_index, generic=False):
        self.J, self.torch, self.variants, self.n_img = J, torch, variants, n_img
        nv = len(variants)
        n_sub = max(1, min(n_sub, n_img))
        self.bounds = [(n_img * s // n_sub, n_img * (s + 1) // n_sub) for s in range(n_sub)]
        flags = J._native.BATCH_EXTERNAL_BUFFERS | (J._native.BATCH_FORCE_GENERIC if generic else 0)
        self.batches = [J.Batch([variants[i % nv]["desc"] for i in range(a, b)], device=device_index, flags=flags) for a, b in self.bounds]
        self.coef_base, self.out_base = [], []
        co = oo = 0
        for b in self.batches:
            self.coef_base.append(co)
            self.out_base.append(oo)
            co += -(-b.coef_arena_bytes() // 256) * 256
            oo += -(-b.out_arena_bytes() // 256) * 256
        self.dev = torch.device("cuda", device_index)
        self.coef_arena = torch.zeros(co, dtype=torch.uint8, device=self.dev)
        self.out_arena = torch.zeros(oo, dtype=torch.uint8, device=self.dev)
        # N distinct coefficient buffers in HBM (no aliasing): every variant goes up once, then ONE broadcast copy per launch
        # group replicates the group's first period of `nv` images over the rest (images of one kind are laid out at a constant
        # stride; a Python loop of one small copy per image and component was 6,144 copies at N = 2 before anything was timed)
        srcs = [[torch.from_numpy(c.view(np.uint8)).to(self.dev) for c in v["coefs"]] for v in variants]
        self.fill_copies = 0
        for s, (a, b) in enumerate(self.bounds):
            bt, base, n = self.batches[s], self.coef_base[s], b - a
            period = None
            if n >= 2 * nv and n % nv == 0 and a % nv == 0:
                period = bt.coef_offset(nv, 0) - bt.coef_offset(0, 0)
                ok = all(bt.coef_offset(i + nv, c) - bt.coef_offset(i, c) == period
                         for i in (0, nv - 1, n - 2 * nv, n - nv - 1) for c in range(len(variants[i % nv]["coefs"])))
                period = period if ok and period > 0 else None
            for i in range(a, a + nv if period else b):
                for c, src in enumerate(srcs[i % nv]):
                    off = base + bt.coef_offset(i - a, c)
                    self.coef_arena[off: off + src.numel()] = src
                    self.fill_copies += 1
            if period:
                first = base + bt.coef_offset(0, 0)
                view = self.coef_arena[first: first + period * (n // nv)].view(n // nv, period)
                view[1:] = view[0:1]
                self.fill_copies += 1
        for s, b in enumerate(self.batches):
            b.bind(self.coef_arena.data_ptr() + self.coef_base[s], self.out_arena.data_ptr() + self.out_base[s])
        self.set_classes(None)

    def set_classes(self, cap):
        """Hand the host-side classification of the coefficients to the batches (what jpgpu_batch_upload computes
        when it stages the data itself); `cap` limits the class (A/B of the arithmetic variants)."""
        nv = len(self.variants)
        for (a, b), batch in zip(self.bounds, self.batches):
            for i in range(a, b):
                cls = self.variants[i % nv]["sane"]
                batch.set_range_hint(i - a, cls if cap is None else min(cls, cap))

    def pixel_slice(self, s):
        b = self.batches[s]
        n = b.n_images
        return self.out_arena[self.out_base[s]: self.out_base[s] + b.out_offset(n - 1) + b.out_bytes(n - 1)]

    def image_pixels(self, i):
        s = next(k for k, (a, b) in enumerate(self.bounds) if a <= i < b)
        b = self.batches[s]
        off = self.out_base[s] + b.out_offset(i - self.bounds[s][0])
        return self.out_arena[off: off + b.out_bytes(i - self.bounds[s][0])]

    def decode(self, stream):
        for b in self.batches:
            b.decode(stream)

    @property
    def path(self):
        return self.batches[0].path

    def close(self):
        for b in self.batches:
            b.close()


def build_variants(J, synth, w, h, sampling, mode, ct, dct_scale=8):
    lum, chr_ = synth.quality_tables(85)
    rgb = synth.synthetic_rgb(w, h)
    ow, oh = (w, h) if dct_scale == 8 else J.scaled_output_size(w, h, dct_scale)
    specs = [([(1, 1), (1, 1), (1, 1)], "ycbcr", "YCbCr"), ([(1, 1)], "gray", "Grayscale")] if mode == "mixed" else [(sampling, mode, ct)]
    variants = []
    for v_sampling, v_mode, v_ct in specs:
        comps, _mcu = J.make_components(w, h, v_sampling, dct_scale=dct_scale)
        qts = [lum] * 4 if v_mode == "cmyk" else ([lum, chr_, chr_, lum] if v_mode == "ycck" else [lum, chr_, chr_][: len(v_sampling)])
        coefs = synth.coefficients_from_rgb(rgb, comps, v_mode, qts)
        # range class of the dequantized coefficients (what jpgpu_batch_upload computes when it stages data itself)
        prod = [np.abs(c.astype(np.int64).reshape(-1, 8, 8) * q.astype(np.int64).reshape(8, 8)) for c, q in zip(coefs, qts)]
        sane = 0
        if all((p < (1 << 15)).all() for p in prod):
            sane = 3 if all((p.sum(axis=1) <= 5900).all() for p in prod) else 1
        variants.append({"sampling": v_sampling, "ct": v_ct, "comps": comps, "qts": qts, "coefs": coefs, "sane": sane,
                         "dct_scale": dct_scale, "desc": J.image_desc(list(comps), qts, ow, oh, v_ct)})
    return variants


class PixelGather:
    """north_star's only collective: the pixels of every rank end on rank 0.  Point-to-point form: each
    peer sends its sub-batch to the root over its own xGMI link while it decodes the next one; nothing to reduce, no ring."""

    def __init__(self, dist, torch, rank, world, slices, device):
        self.dist, self.rank, self.world = dist, rank, world
        self.slices = slices
        # shards may differ by one image: every rank tells the others how many bytes its sub-batches hold
        mine = torch.tensor([sl.numel() for sl in slices], dtype=torch.int64, device=device)
        every = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        self.sizes = [[int(x) for x in t.cpu()] for t in every]
        if any(len(t) != len(slices) for t in self.sizes):
            raise SystemExit("bench.py: ranks disagree on the number of sub-batches")
        self.recv = None
        if rank == 0:
            self.recv = [[torch.empty(n, dtype=torch.uint8, device=device) for n in self.sizes[r]] for r in range(1, world)]
        self.pending = []

    def post(self, s):
        """Enqueue the transfers of sub-batch `s` behind what the current stream has been given so far."""
        P = self.dist.P2POp
        if self.rank == 0:
            ops = [P(self.dist.irecv, self.recv[r - 1][s], r) for r in range(1, self.world)]
        else:
            ops = [P(self.dist.isend, self.slices[s], 0)]
        if ops:
            self.pending += self.dist.batch_isend_irecv(ops)

    def wait(self):
        for wk in self.pending:
            wk.wait()
        self.pending = []
