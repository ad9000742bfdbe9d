"""TEST INFRASTRUCTURE -- generates tests/golden/input_pil.npz: ground truth for the tail of the reference's input pipeline
(utils.py:122-129: RandomHorizontalFlip -> RandomCrop -> ToTensor -> Normalize) from an implementation INDEPENDENT of both
oracle/input_oracle.py and the device kernel: the flip and the crop are done by PIL itself (what torchvision calls for PIL
images: Image.transpose(FLIP_LEFT_RIGHT), Image.crop((left, top, left + w, top + h))), ToTensor / Normalize by NumPy in
float32 (/ 255, then (t - 0.5) / 0.5, HWC -> CHW).  torchvision is not installed in this image, so this is the closest
available pin; the fixture is committed and travels to the GPU box.

    python oracle/make_input_golden.py"""
import os

import numpy as np
from PIL import Image

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "input_pil.npz")


def main():
    rng = np.random.RandomState(20240926)
    Hs, Ws, H, W, N = 71, 83, 64, 64, 6
    imgs = rng.randint(0, 256, size=(N, Hs, Ws, 3)).astype(np.uint8)
    imgs[0, :2, :2] = [[[0, 255, 1], [254, 127, 128]], [[3, 85, 170], [17, 34, 51]]]      # range ends and thirds
    tops = rng.randint(0, Hs - H + 1, size=N)
    lefts = rng.randint(0, Ws - W + 1, size=N)           # crop window in the (possibly flipped) image, as RandomCrop draws it
    flips = np.array([0, 1, 1, 0, 1, 0], dtype=bool)
    outs = []
    for n in range(N):
        im = Image.fromarray(imgs[n], mode="RGB")
        if flips[n]:
            im = im.transpose(Image.FLIP_LEFT_RIGHT)
        im = im.crop((int(lefts[n]), int(tops[n]), int(lefts[n]) + W, int(tops[n]) + H))
        a = np.asarray(im, dtype=np.uint8)
        t = a.astype(np.float32) / np.float32(255)                      # ToTensor
        t = (t - np.float32(0.5)) / np.float32(0.5)                     # Normalize((0.5,)*3, (0.5,)*3)
        outs.append(np.ascontiguousarray(t.transpose(2, 0, 1)))         # HWC -> CHW
    np.savez_compressed(OUT, images=imgs, tops=tops, lefts=lefts, flips=flips, height=H, width=W, out=np.stack(outs))
    print("wrote", OUT, np.stack(outs).shape)


if __name__ == "__main__":
    main()
