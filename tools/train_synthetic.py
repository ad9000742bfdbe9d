"""The reference's training loop (train.py:226-262, 348-388) driven through this package's drop-in API, on synthetic images:
what a user of Onr/Council-GAN changes is the import of `Council_Trainer` and -- optionally -- the input tail
(`DeviceInput` instead of ToTensor/Normalize + a blocking `.cuda()`).

    python tools/train_synthetic.py --config configs/male2female_council_folder.yaml --iterations 20 --output out/ [--resume]

Writes the reference's checkpoint files into <output>/checkpoints at `snapshot_save_iter` and at the end, and the 8-tuple of
`sample()` as PNG strips into <output>/images at `image_save_iter`."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from PIL import Image  # noqa: E402

import council_gan_amd as cga  # noqa: E402


def synthetic_u8(rng, batch, size):
    """Stand-in for the loader: decoded (already resized) uint8 HWC images of both domains."""
    return (rng.randint(0, 256, size=(batch, size, size, 3)).astype(np.uint8),
            rng.randint(0, 256, size=(batch, size, size, 3)).astype(np.uint8))


def save_strip(tensors, path):
    rows = [t for t in tensors if t is not None]
    if not rows:
        return
    img = torch.cat([torch.cat(list(t.float().clamp(-1, 1)), 2) for t in rows], 1)      # rows = outputs, columns = samples
    arr = ((img + 1) * 127.5 + 0.5).clamp(0, 255).permute(1, 2, 0).to(torch.uint8).cpu().numpy()
    Image.fromarray(arr).save(path)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', required=True)
    ap.add_argument('--output', required=True)
    ap.add_argument('--iterations', type=int, default=10)
    ap.add_argument('--resume', action='store_true')
    ap.add_argument('--device', default='cuda:0')
    args = ap.parse_args(argv)

    config = cga.get_config(args.config)
    cga.seed_everything(config['random_seed'])                     # train.py:55-62
    cga.init_distributed()                                         # no-op in a single process
    trainer = cga.Council_Trainer(config, args.device)
    trainer.cuda(args.device)
    ck_dir, im_dir = os.path.join(args.output, 'checkpoints'), os.path.join(args.output, 'images')
    os.makedirs(ck_dir, exist_ok=True)
    os.makedirs(im_dir, exist_ok=True)
    iterations = trainer.resume(ck_dir, hyperparameters=config) if args.resume else 0      # train.py:103
    pipe = cga.DeviceInput(args.device, config['crop_image_height'], config['crop_image_width'])
    rng = np.random.RandomState(config['random_seed'])
    size = max(config['new_size'], config['crop_image_height'], config['crop_image_width'])
    n_dis = max(config['dis']['numberOf_dis_relative_iteration'], 1)
    b = config['batch_size']
    last = iterations + args.iterations
    dis_iter = 1
    while iterations < last:
        u8_a, u8_b = synthetic_u8(rng, b, size)
        tl = np.stack([rng.randint(0, size - config['crop_image_height'] + 1, size=b),
                       rng.randint(0, size - config['crop_image_width'] + 1, size=b)], 1)     # RandomCrop windows
        images_a, images_b = pipe(u8_a, crop_tl=tl), pipe(u8_b, crop_tl=tl)
        t = time.time()
        config['iteration'] = iterations
        trainer.dis_update(images_a, images_b, config)                                          # train.py:244-251
        if dis_iter < n_dis:
            dis_iter += 1
            continue
        dis_iter = 1
        if config['council']['numberOfCouncil_dis_relative_iteration'] > 0:
            trainer.dis_council_update(images_a, images_b, config)
        trainer.gen_update(images_a, images_b, config, iterations)
        trainer.update_learning_rate()
        torch.cuda.synchronize()
        iterations += 1
        print("iteration %d  %.1f ms  loss_gen %s  loss_dis %s" % (
            iterations, 1e3 * (time.time() - t), ["%.3f" % float(v) for v in trainer.loss_gen_total_s],
            ["%.3f" % float(v) for v in trainer.loss_dis_total_s]))
        if iterations % config['image_save_iter'] == 0 or iterations == last:
            n = min(config['display_size'], b)
            save_strip(trainer.sample(images_a[:n], images_b[:n]), os.path.join(im_dir, 'sample_%08d.png' % iterations))
        if iterations % config['snapshot_save_iter'] == 0 or iterations == last:
            trainer.save(ck_dir, iterations - 1)                  # files are named with iterations (train.py:388)
    return iterations


if __name__ == "__main__":
    main()
