"""Host-side glue the trainer constructor relies on (reference utils.py:183-185, 336-348, 392-422)."""
import math
import os

import torch.nn.init as init
import yaml
from torch.optim import lr_scheduler


def get_config(config):
    """utils.py:183-185 -- the YAML schema is API surface and is parsed unchanged."""
    with open(config, 'r') as stream:
        return yaml.safe_load(stream)


def weights_init(init_type='gaussian'):
    """utils.py:402-422: applies to every module whose class name starts with Conv / Linear and
    that owns a `weight` (i.e. nn.Conv2d / nn.Linear; Conv2dBlock has none)."""
    def init_fun(m):
        classname = m.__class__.__name__
        if (classname.find('Conv') == 0 or classname.find('Linear') == 0) and hasattr(m, 'weight'):
            if init_type == 'gaussian':
                init.normal_(m.weight.data, 0.0, 0.02)
            elif init_type == 'xavier':
                init.xavier_normal_(m.weight.data, gain=math.sqrt(2))
            elif init_type == 'kaiming':
                init.kaiming_normal_(m.weight.data, a=0, mode='fan_in')
            elif init_type == 'orthogonal':
                init.orthogonal_(m.weight.data, gain=math.sqrt(2))
            elif init_type == 'default':
                pass
            else:
                assert 0, "Unsupported initialization: {}".format(init_type)
            if hasattr(m, 'bias') and m.bias is not None:
                init.constant_(m.bias.data, 0.0)
    return init_fun


def get_scheduler(optimizer, hyperparameters, iterations=-1):
    """utils.py:392-400."""
    if 'lr_policy' not in hyperparameters or hyperparameters['lr_policy'] == 'constant':
        return None
    if hyperparameters['lr_policy'] == 'step':
        if iterations != -1:
            for g in optimizer.param_groups:       # StepLR(last_epoch != -1) requires initial_lr
                g.setdefault('initial_lr', hyperparameters['lr'])
        return lr_scheduler.StepLR(optimizer, step_size=hyperparameters['step_size'],
                                   gamma=hyperparameters['gamma'], last_epoch=iterations)
    raise NotImplementedError('learning rate policy [%s] is not implemented' % hyperparameters['lr_policy'])


def get_model_list(dirname, key):
    """utils.py:336-348: lexicographically last checkpoint whose name contains `key`."""
    if os.path.exists(dirname) is False:
        return None
    models = [os.path.join(dirname, f) for f in os.listdir(dirname)
              if os.path.isfile(os.path.join(dirname, f)) and key in f and ".pt" in f]
    models.sort()
    return models[-1] if models else None


def seed_everything(seed=1):
    """train.py:55-62: Python, NumPy and torch generators from one seed (identical on every rank)."""
    import random
    import numpy as np
    import torch
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


def synthetic_batch(batch, size, seed=7):
    """Benchmark input (no dataset in the image): fp32 images in [-1, 1), the range Normalize(0.5, 0.5) produces
    (utils.py:124-126); both domains from one seeded generator."""
    import torch
    g = torch.Generator().manual_seed(seed)
    x_a = torch.rand(batch, 3, size, size, generator=g) * 2 - 1
    x_b = torch.rand(batch, 3, size, size, generator=g) * 2 - 1
    return x_a, x_b
