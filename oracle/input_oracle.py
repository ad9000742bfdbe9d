"""TEST INFRASTRUCTURE -- CPU restatement of the tail of the reference's input pipeline (utils.py:124-129,133-134:
`RandomHorizontalFlip`, `RandomCrop((height, width))`, `ToTensor()`, `Normalize((0.5,)*3, (0.5,)*3)`), for the parity
test of council_gan_amd/input.py.  Only tests/ may import this module.

Parity unpinned against torchvision itself: the reference's loader needs torchvision, which is not installed in this
image.  The four transforms are restated from their published definitions (torchvision.transforms.functional):
  hflip(img)            = img[:, ::-1]                                   (PIL FLIP_LEFT_RIGHT)
  crop(img, i, j, h, w) = img[i:i+h, j:j+w]
  to_tensor(img)        = from_numpy(img).permute(2, 0, 1).contiguous().to(float32).div(255)
  normalize(t, m, s)    = t.sub_(m).div_(s)                              (fp32, per channel)."""
import numpy as np
import torch


def to_tensor_normalize(img_u8, mean=0.5, std=0.5):
    t = torch.from_numpy(np.ascontiguousarray(img_u8)).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
    return t.sub_(mean).div_(std)


def sample(img_u8, top, left, height, width, flip_first=False, mean=0.5, std=0.5):
    """One sample in the reference's order: (flip the whole image) -> crop window -> ToTensor -> Normalize."""
    img = np.asarray(img_u8)
    if flip_first:
        img = img[:, ::-1]
    return to_tensor_normalize(img[top:top + height, left:left + width], mean, std)


def window_after_flip(left_in_flipped, image_width, width):
    """Cropping the flipped image at `left` equals flipping the crop of the ORIGINAL image at this left."""
    return image_width - width - left_in_flipped
