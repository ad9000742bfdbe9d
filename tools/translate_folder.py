"""Translate a folder of images with the council's generators (the job of the reference's test_on_folder.py:61-167, on
the MI355X path): every image -> one council member's generator (or all of them) x `--num_style` random style codes.

    python tools/translate_folder.py --config configs/male2female_council_folder.yaml --checkpoint_dir outputs/.../checkpoints \
        --input_folder imgs/ --output_folder out/ [--a2b 1] [--num_style 4] [--member -1 | i | all] [--seed 1]

The checkpoint directory holds the reference's files (`a2b_gen_{i}_{iteration:08d}.pt`, trainer_council.py:969-992) -- the
authors' published `pretrain/*` checkpoints included; without `--checkpoint_dir` the generators keep their initial weights
(smoke runs).  Images are resized so that the shorter side is `new_size` (PIL bilinear, as transforms.Resize) and centre-cropped to
the configured crop; the uint8 -> normalised NHWC conversion runs on the device (council_gan_amd.DeviceInput).  Outputs are
written as PNG after the per-image min-max normalisation torchvision.utils.save_image(normalize=True) applies."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from PIL import Image  # noqa: E402

import council_gan_amd as cga  # noqa: E402

EXTS = ('.jpg', '.jpeg', '.png', '.ppm', '.bmp', '.webp')


def load_image(path, new_size, height, width):
    """uint8 [height, width, 3]: shorter side -> new_size (bilinear), then the centre crop."""
    im = Image.open(path).convert('RGB')
    w, h = im.size
    if new_size:
        if w <= h:
            im = im.resize((new_size, max(1, int(new_size * h / w))), Image.BILINEAR)
        else:
            im = im.resize((max(1, int(new_size * w / h)), new_size), Image.BILINEAR)
    w, h = im.size
    if w < width or h < height:
        raise ValueError("%s: %dx%d after resizing is smaller than the %dx%d crop" % (path, w, h, width, height))
    left, top = int(round((w - width) / 2.0)), int(round((h - height) / 2.0))
    return np.array(im.crop((left, top, left + width, top + height)), dtype=np.uint8)      # a writable copy


def save_normalised(t, path):
    """t: [1, 3, H, W] fp32 on the device -> PNG, scaled so that min -> 0 and max -> 255."""
    t = t[0].float()
    lo, hi = float(t.min()), float(t.max())
    t = (t - lo) / max(hi - lo, 1e-5)
    arr = (t.clamp(0, 1) * 255.0 + 0.5).permute(1, 2, 0).to(torch.uint8).cpu().numpy()
    Image.fromarray(arr).save(path)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('--config', required=True)
    ap.add_argument('--input_folder', required=True)
    ap.add_argument('--output_folder', required=True)
    ap.add_argument('--checkpoint_dir', default=None)
    ap.add_argument('--a2b', type=int, default=1, help="1: domain a -> b, 0: b -> a")
    ap.add_argument('--num_style', type=int, default=4)
    ap.add_argument('--member', default='-1', help="-1: a random member per image (as the reference), i: member i, all: every member")
    ap.add_argument('--seed', type=int, default=1)
    ap.add_argument('--device', default='cuda:0')
    args = ap.parse_args(argv)

    cfg = cga.get_config(args.config)
    d = 'a2b' if args.a2b else 'b2a'
    if not cfg['do_' + d]:
        raise SystemExit("the configuration does not train the %s direction" % d)
    cga.seed_everything(args.seed)
    trainer = cga.Council_Trainer(cfg, args.device)
    trainer.cuda(args.device)
    if args.checkpoint_dir:
        it = trainer.resume(args.checkpoint_dir, cfg)
        print("generators of iteration %d" % it)
    trainer.eval()
    gens = trainer.gen_a2b_s if args.a2b else trainer.gen_b2a_s
    council = len(gens)
    pipe = cga.DeviceInput(args.device, cfg['crop_image_height'], cfg['crop_image_width'])
    new_size = cfg.get('new_size_a' if args.a2b else 'new_size_b', cfg.get('new_size'))
    names = sorted(f for f in os.listdir(args.input_folder) if f.lower().endswith(EXTS))
    os.makedirs(args.output_folder, exist_ok=True)
    rng = np.random.RandomState(args.seed)
    written = []
    with torch.no_grad():
        for n, name in enumerate(names):
            img = load_image(os.path.join(args.input_folder, name), new_size, cfg['crop_image_height'], cfg['crop_image_width'])
            x = pipe(img[None])
            members = range(council) if args.member == 'all' else [rng.randint(council) if int(args.member) < 0 else int(args.member)]
            torch.manual_seed(args.seed + 1 + n)             # the reference re-seeds per image (test_on_folder.py:129-131)
            styles = torch.randn(args.num_style, cfg['gen']['style_dim'], 1, 1).to(args.device)
            for k in members:
                content, _ = gens[k].encode(x)
                for j in range(args.num_style):
                    out = gens[k].decode(content, styles[j:j + 1], x)
                    path = os.path.join(args.output_folder, "%s_m%d_s%02d.png" % (os.path.splitext(name)[0], k, j))
                    save_normalised(out, path)
                    written.append(path)
    print("wrote %d images to %s" % (len(written), args.output_folder))
    return written


if __name__ == "__main__":
    main()
