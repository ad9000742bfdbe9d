"""Input pipeline tail on the device.

The reference's loader (utils.py:122-181) ends every sample with `RandomCrop((H, W))` (train) -> `ToTensor()` ->
`Normalize(0.5, 0.5)` on the host and train.py:228 then uploads fp32 NCHW batches with a blocking `.cuda()`.  Here the
host hands over the decoded (and PIL-resized / PIL-augmented) images as uint8 HWC -- a quarter of the bytes -- through a
ring of pinned staging buffers and a non-blocking copy, and one HIP kernel applies the crop window (and, for loaders
that defer it, the horizontal flip), the /255 and the normalisation, writing the NHWC fp32 batch the trainer consumes
directly (`Council_Trainer._img` then neither copies nor re-lays it out).

The PIL-space transforms (Resize and the optional colour / affine augmentations) stay with the host loader: their
results are defined by PIL's filters, not by arithmetic this path could restate."""
import torch

from . import hip
from .hip import check, ptr, stream


class DeviceInput:
    def __init__(self, device, height, width, mean=0.5, std=0.5, ring=4):
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise hip.HipError("DeviceInput runs on an MI355X only (got device %s)" % self.device)
        self.H, self.W, self.mean, self.std = int(height), int(width), float(mean), float(std)
        self._ring = [None] * ring
        self._i = 0

    def _stage(self, t):
        """Pinned staging buffer + non-blocking H2D copy; a slot is reused only after its previous copy has completed."""
        k = self._i % len(self._ring)
        self._i += 1
        slot = self._ring[k]
        n = t.numel() * t.element_size()
        if slot is None or slot[0].numel() < n:
            slot = [torch.empty(max(n, 1 << 20), dtype=torch.uint8, pin_memory=True), None]
            self._ring[k] = slot
        if slot[1] is not None:
            slot[1].synchronize()
        host = slot[0][:n].view(t.dtype).view(t.shape)
        host.copy_(t)
        dev = host.to(self.device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        slot[1] = ev
        return dev

    def __call__(self, images_u8, crop_tl=None, flip=None):
        """images_u8: uint8 [N, Hs, Ws, C] (host tensor or array, HWC as PIL decodes it); crop_tl: [N, 2] ints
        (top, left) of each sample's (H, W) window, None = top-left corner; flip: [N] bools, None = no flip.
        Returns the fp32 batch as a logical [N, C, H, W] tensor in channels_last (physical NHWC) layout."""
        x = torch.as_tensor(images_u8)
        if x.dtype != torch.uint8 or x.dim() != 4:
            raise ValueError("expected a uint8 [N, Hs, Ws, C] batch, got %s %s" % (x.dtype, tuple(x.shape)))
        N, Hs, Ws, C = x.shape
        if self.H > Hs or self.W > Ws:
            raise ValueError("crop window %dx%d larger than the %dx%d images" % (self.H, self.W, Hs, Ws))
        xd = x if x.is_cuda else self._stage(x.contiguous())
        cd = fd = None
        if crop_tl is not None:
            c = torch.as_tensor(crop_tl, dtype=torch.int32).reshape(N, 2)
            if int(c[:, 0].min()) < 0 or int(c[:, 1].min()) < 0 or int(c[:, 0].max()) > Hs - self.H or int(c[:, 1].max()) > Ws - self.W:
                raise ValueError("crop window outside the image")
            cd = c if c.is_cuda else self._stage(c.contiguous())
        if flip is not None:
            f = torch.as_tensor(flip).reshape(N).to(torch.uint8)
            fd = f if f.is_cuda else self._stage(f.contiguous())
        out = torch.empty((N, C, self.H, self.W), dtype=torch.float32, device=self.device,
                          memory_format=torch.channels_last)
        check(hip.load().cg_u8_to_f32_nhwc(ptr(xd), N, Hs, Ws, C, ptr(cd), ptr(fd), self.H, self.W, self.mean, self.std,
                                           ptr(out), stream()), "cg_u8_to_f32_nhwc")
        return out
