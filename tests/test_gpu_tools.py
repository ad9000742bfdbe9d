"""tools/translate_folder.py end to end: checkpoint written by save() -> folder of PNGs -> translated PNGs, and the
translation equals a direct encode/decode of the same image with the same style code."""
import copy
import os
import sys

import numpy as np
import pytest
import torch
import yaml
from PIL import Image

pytestmark = pytest.mark.gpu


def test_translate_folder_cli(tmp_path):
    import council_gan_amd as cga
    root = os.path.join(os.path.dirname(__file__), "..")
    sys.path.insert(0, os.path.join(root, "tools"))
    import translate_folder as TF
    cfg = yaml.safe_load(open(os.path.join(root, "configs", "male2female_council_folder.yaml")))
    cfg['gen'].update(dim=16, mlp_dim=32, n_res=2)
    cfg['dis'].update(dim=16)
    cfg['council']['council_size'] = 2
    cfg['new_size'] = 72
    cfg['crop_image_height'] = cfg['crop_image_width'] = 64
    cfg_path = tmp_path / "cfg.yaml"
    yaml.safe_dump(cfg, open(cfg_path, "w"))
    cga.seed_everything(5)
    tr = cga.Council_Trainer(copy.deepcopy(cfg), 'cuda:0')
    tr.cuda('cuda:0')
    ck = tmp_path / "ck"
    ck.mkdir()
    tr.save(str(ck), 41)
    src = tmp_path / "in"
    src.mkdir()
    rng = np.random.RandomState(0)
    for i, (w, h) in enumerate(((90, 120), (150, 100))):
        Image.fromarray(rng.randint(0, 256, size=(h, w, 3)).astype(np.uint8)).save(src / ("img%d.png" % i))
    out = tmp_path / "out"
    written = TF.main(["--config", str(cfg_path), "--checkpoint_dir", str(ck), "--input_folder", str(src),
                       "--output_folder", str(out), "--num_style", "2", "--member", "all", "--seed", "3"])
    assert len(written) == 2 * 2 * 2 and all(os.path.exists(p) for p in written)
    im = np.asarray(Image.open(written[0]))
    assert im.shape == (64, 64, 3) and im.dtype == np.uint8 and im.max() == 255 and im.min() == 0
    # the first written file = image 0, member 0, style 0: reproduce it directly
    img = TF.load_image(str(src / "img0.png"), 72, 64, 64)
    assert img.shape == (64, 64, 3)
    x = cga.DeviceInput('cuda:0', 64, 64)(img[None])
    torch.manual_seed(3 + 1 + 0)
    s = torch.randn(2, cfg['gen']['style_dim'], 1, 1).cuda()
    tr.eval()
    with torch.no_grad():
        c, _ = tr.gen_a2b_s[0].encode(x)
        ref = tr.gen_a2b_s[0].decode(c, s[0:1], x)[0]
    lo, hi = float(ref.min()), float(ref.max())
    want = (((ref - lo) / (hi - lo)).clamp(0, 1) * 255.0 + 0.5).permute(1, 2, 0).to(torch.uint8).cpu().numpy()
    assert np.abs(want.astype(int) - im.astype(int)).max() <= 1


def test_train_synthetic_loop_and_resume(tmp_path):
    """tools/train_synthetic.py: the reference's loop order through the drop-in API (device input tail, three updates,
    scheduler step, sample strip, checkpoint), then a resumed run continuing the iteration count."""
    root = os.path.join(os.path.dirname(__file__), "..")
    sys.path.insert(0, os.path.join(root, "tools"))
    import train_synthetic as TS
    cfg = yaml.safe_load(open(os.path.join(root, "configs", "male2female_council_folder.yaml")))
    cfg['gen'].update(dim=16, mlp_dim=32, n_res=2)
    cfg['dis'].update(dim=16)
    cfg['council']['council_size'] = 2
    cfg['batch_size'] = 2
    cfg['new_size'] = 72
    cfg['crop_image_height'] = cfg['crop_image_width'] = 64
    cfg['council']['council_start_at_iter'] = 0
    cfg_path = tmp_path / "cfg.yaml"
    yaml.safe_dump(cfg, open(cfg_path, "w"))
    out = tmp_path / "run"
    assert TS.main(["--config", str(cfg_path), "--output", str(out), "--iterations", "2"]) == 2
    files = sorted(os.listdir(out / "checkpoints"))
    assert 'a2b_gen_0_00000002.pt' in files and 'a2b_dis_council_1_00000002.pt' in files and 'optimizer_1.pt' in files
    strip = np.asarray(Image.open(out / "images" / "sample_00000002.png"))
    assert strip.ndim == 3 and strip.shape[2] == 3 and strip.shape[0] % 64 == 0 and strip.shape[1] % 64 == 0
    assert TS.main(["--config", str(cfg_path), "--output", str(out), "--iterations", "1", "--resume"]) == 3
    assert 'a2b_gen_0_00000003.pt' in os.listdir(out / "checkpoints")
