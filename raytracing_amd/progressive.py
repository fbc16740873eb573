"""Progressive renders: a frame that gains samples call by call and can be put away and taken up again.

The reference renders a frame in one go and keeps nothing in between (src/main.c:742-792: the
per-pixel sum lives in a local of raytrace_region).  On the device the running sums are a buffer
of their own (RT_GPU_FLAG_ACCUMULATE: un-normalised sums, read, add, store) and sample i of a
pixel always draws from the stream keyed (seed, pixel, i), so rendering n samples now and m later
gives, bit for bit, the frame of n + m samples in one call.  A checkpoint is the sums plus the
number of samples they hold.
"""
import numpy as np

from . import gpu


class ProgressiveFrame:
    """Running per-pixel sums of a frame on the current device."""

    def __init__(self, gpu_scene, width, height, seed, max_segments=0):
        import torch
        self.scene, self.width, self.height, self.seed = gpu_scene, int(width), int(height), int(seed)
        self.max_segments = int(max_segments)
        self.samples = 0                              # samples per pixel the sums hold
        self.sums = torch.zeros((self.height, self.width, 3), dtype=torch.float32, device="cuda")

    def add_samples(self, count, stream=0):
        """Render samples [self.samples, self.samples + count) of every pixel into the sums."""
        if count <= 0:
            raise ValueError("count must be positive")
        o = gpu.Options(flags=gpu.FLAG_ACCUMULATE, sample_offset=self.samples, max_segments=self.max_segments)
        self.scene.render_region_device(self.sums.data_ptr(), self.width, self.height,
                                        (0, 0, self.width, self.height), count, self.seed, o, stream)
        self.samples += count

    def frame(self):
        """The frame so far: sums scaled by the reference's final factor for the samples held (a new tensor)."""
        if self.samples == 0:
            raise ValueError("no samples yet")
        out = self.sums.clone()
        gpu.normalize_device(out.data_ptr(), self.width * self.height, self.samples)
        return out

    # ---- checkpoint: the sums and what they stand for, in one .npz file ---------------------

    def save(self, path):
        import torch
        torch.cuda.synchronize()
        np.savez(path, sums=self.sums.cpu().numpy(),
                 meta=np.array([self.width, self.height, self.samples, self.seed & 0xFFFFFFFF, self.seed >> 32,
                                self.max_segments], dtype=np.uint64))

    @classmethod
    def load(cls, path, gpu_scene):
        import torch
        with np.load(path, allow_pickle=False) as z:
            sums, meta = z["sums"], [int(v) for v in z["meta"]]
        check_checkpoint(sums, meta)
        f = cls(gpu_scene, meta[0], meta[1], meta[3] | (meta[4] << 32), meta[5])
        f.samples = meta[2]
        f.sums.copy_(torch.from_numpy(sums))
        return f


def check_checkpoint(sums, meta):
    """A checkpoint file must describe its own array (host-side check, no device needed)."""
    if len(meta) != 6:
        raise ValueError("checkpoint: unknown header")
    width, height = meta[0], meta[1]
    if sums.dtype != np.float32 or sums.shape != (height, width, 3):
        raise ValueError(f"checkpoint: sums are {sums.dtype} {sums.shape}, header says {height}x{width}")
    if not np.isfinite(sums).all():
        raise ValueError("checkpoint: non-finite sums")
