"""Networks of the Council-GAN hot path on the gfx950 kernels.

Mirror of the reference's operator API (`/root/reference/networks.py`): the same class names,
constructor arguments, attribute names and module tree, hence the same `state_dict` keys and
shapes (published checkpoints load) and the same RNG consumption at construction (same seed ->
same initial weights).  `nn.Conv2d` / `nn.Linear` objects are kept purely as parameter holders;
their `forward` is never called -- every forward/backward goes through `ops` (HIP kernels).

Layout: activations are channels_last (physical NHWC) fp32.  Fusions relative to the reference's
module-by-module execution:
  * ZeroPad2d is folded into the conv gather; bias + activation into the conv epilogue;
  * InstanceNorm / AdaIN + activation + the ResBlock residual add run as one apply pass;
  * nn.Upsample(2x nearest) is folded into the next conv's gather;
  * torch.cat((x, x_input), 1) of the council discriminator is a two-source gather.
"""
import torch
import torch.nn as nn

from . import hip, ops


def _unsupported(what):
    raise NotImplementedError("%s is not reachable from the shipped configs (SURVEY.md 8a) and is not "
                              "implemented on the HIP path" % what)


##################################################################################
# Normalization layers
##################################################################################
class AdaptiveInstanceNorm2d(nn.Module):
    """networks.py:627-656.  weight/bias are assigned per forward by AdaINGen.assign_adain_params;
    here they are column ranges of the MLP output, consumed in place by the AdaIN kernel."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1):
        super().__init__()
        self.num_features = num_features
        self.eps = eps
        self.momentum = momentum
        self.weight = None
        self.bias = None
        self.params = None      # [B, P] MLP output
        self.pgrad = None       # ops.ParamGrad of that matrix (shared by the AdaIN layers of one decoder pass) or None
        self.goff = self.boff = 0
        # dummy buffers, kept for state_dict compatibility (networks.py:636-638)
        self.register_buffer('running_mean', torch.zeros(num_features))
        self.register_buffer('running_var', torch.ones(num_features))

    def forward(self, x, act='none', residual=None, stats=None, want_split=False, want_f32=True, skip=None):
        assert self.params is not None, "Please assign weight and bias before calling AdaIN!"
        return ops.adain(x, self.params, self.goff, self.boff, act=act, residual=residual, eps=self.eps, stats=stats,
                         want_split=want_split, want_f32=want_f32, skip=skip, pgrad=self.pgrad)

    def __repr__(self):
        return self.__class__.__name__ + '(' + str(self.num_features) + ')'


class LayerNorm(nn.Module):
    """networks.py:659-686."""

    def __init__(self, num_features, eps=1e-5, affine=True):
        super().__init__()
        self.num_features = num_features
        self.affine = affine
        self.eps = eps
        if self.affine:
            self.gamma = nn.Parameter(torch.Tensor(num_features).uniform_())
            self.beta = nn.Parameter(torch.zeros(num_features))

    def forward(self, x):
        if not self.affine:
            _unsupported("LayerNorm(affine=False)")
        if x.dim() != 4:
            _unsupported("LayerNorm on non-4D input")
        return ops.layer_norm(x, self.gamma, self.beta, self.eps)


##################################################################################
# Basic Blocks
##################################################################################
class Conv2dBlock(nn.Module):
    """networks.py:463-521: ZeroPad2d -> Conv2d(bias) -> norm -> activation, as one or two kernels."""

    def __init__(self, input_dim, output_dim, kernel_size, stride,
                 padding=0, norm='none', activation='relu', pad_type='zero'):
        super().__init__()
        self.use_bias = True
        if pad_type != 'zero':
            _unsupported("pad_type %r" % pad_type)
        self.padding = padding
        self.stride = stride
        self.kernel_size = kernel_size
        norm_dim = output_dim
        self.norm_type = norm
        if norm == 'in':
            self.norm = nn.InstanceNorm2d(norm_dim)      # stateless marker (affine=False): no parameters
        elif norm == 'ln':
            self.norm = LayerNorm(norm_dim)
        elif norm == 'adain':
            self.norm = AdaptiveInstanceNorm2d(norm_dim)
        elif norm == 'none':
            self.norm = None
        else:
            _unsupported("norm %r" % norm)
        if activation not in ('relu', 'lrelu', 'tanh', 'none'):
            _unsupported("activation %r" % activation)
        self.activation_type = activation
        self.conv = nn.Conv2d(input_dim, output_dim, kernel_size, stride, bias=self.use_bias)

    def forward(self, x, x2=None, upsample=False, residual=None, want_f32=True, skip=None):
        # skip (ops.SkipLink, ResBlock only): the block's second layer (residual given) hands the skip edge's gradient over in
        # its norm's backward, the first layer (no residual) adds it to its convolution's data gradient
        # want_f32=False: the caller knows that the only consumer of this block's output is a convolution that reads the
        # {hi, lo} planes the norm emits -- the fp32 copy is then not written (split-precision datapath only)
        act = self.activation_type
        fused_act = act if self.norm is None else 'none'
        stats = [] if self.norm_type in ('in', 'adain') else None     # conv epilogue -> norm statistics hand-off
        # split-precision forward (ops.conv2d): the layer's output also leaves in {hi, lo} fp16 form when the next
        # convolution can consume it (its input width = this layer's output width)
        wmgr = getattr(self, '_cg_wmgr', None)
        want_split = wmgr is not None and self.conv.out_channels % 32 == 0
        y = ops.conv2d(x, self.conv.weight, self.conv.bias, self.stride, self.padding, fused_act, x2=x2,
                       upsample=upsample, stats=stats, wmgr=wmgr, want_split=want_split,
                       want_f32=want_f32 or self.norm is not None, skip=skip if residual is None else None)
        if self.norm_type == 'in':
            y = ops.instance_norm(y, act=act, residual=residual, eps=self.norm.eps, stats=stats, want_split=want_split,
                                  want_f32=want_f32, skip=skip)
        elif self.norm_type == 'adain':
            y = self.norm(y, act=act, residual=residual, stats=stats, want_split=want_split, want_f32=want_f32, skip=skip)
        elif self.norm_type == 'ln':
            y = ops.activation(self.norm(y), act)
            if residual is not None:
                y = y + residual
        elif residual is not None:
            y = y + residual
        return y


def _split_ok(block):
    """A Conv2dBlock the split-precision path can run: instance-norm / AdaIN after a conv whose input width is a
    multiple of 32, with its weight registered in a split-weight table (Council_Trainer refreshes it)."""
    co = block.conv.out_channels
    return (getattr(block, '_cg_wsplit', None) is not None and block.norm_type in ('in', 'adain')
            and block.conv.in_channels % 32 == 0
            and co % 32 == 0 and 256 % (co // 4) == 0)      # cg_instnorm_apply_split's channel-quad layout


def conv_block_split(block, xs, upsample=False, residual=None, want_f32=False):
    """Conv2dBlock.forward on split-precision operands (tape-free passes only): returns (y_fp32_or_None, SplitTensor)."""
    k = block.kernel_size
    stats = []
    wmgr = getattr(block, '_cg_wmgr', None)
    if upsample and ops._upconv_ok(tuple(xs.shape), block.conv.weight, block.stride, block.padding, 'none', None, wmgr):
        # nearest-2x upsample + 3x3: the summed-tap transposed-convolution form (2.25x fewer multiply-adds)
        y = ops.upconv_fwd_x3(xs, block.conv.weight, block.conv.bias, wmgr, ops._grp(block.conv.weight), ops.group_n(), stats)
    else:
        y = ops.conv2d_x3(xs, block._cg_wsplit, block.conv.out_channels, k, k, block.conv.bias, block.stride, block.padding,
                          'none', upsample=upsample, stats=stats, grp=ops._grp(block.conv.weight))
    if block.norm_type == 'adain':
        n = block.norm
        assert n.params is not None, "Please assign weight and bias before calling AdaIN!"
        return ops.instnorm_split(y, n.params, n.goff, n.boff, act=block.activation_type, residual=residual, eps=n.eps,
                                  stats=stats, want_f32=want_f32)
    return ops.instnorm_split(y, None, 0, 0, act=block.activation_type, residual=residual, eps=block.norm.eps,
                              stats=stats, want_f32=want_f32)


class ResBlock(nn.Module):
    """networks.py:448-461; `out += residual` is fused into the second norm's apply pass."""

    def __init__(self, dim, norm='in', activation='relu', pad_type='zero'):
        super().__init__()
        model = []
        model += [Conv2dBlock(dim, dim, 3, 1, 1, norm=norm, activation=activation, pad_type=pad_type)]
        model += [Conv2dBlock(dim, dim, 3, 1, 1, norm=norm, activation='none', pad_type=pad_type)]
        self.model = nn.Sequential(*model)

    def forward(self, x):
        # the skip edge's gradient joins the first convolution's data gradient (ops.SkipLink) when both layers have a norm whose
        # backward can hand it over; otherwise the autograd engine adds the two gradients of x as usual
        link = ops.skip_link(x) if self.model[1].norm_type in ('in', 'adain') else None
        return self.model[1](self.model[0](x, want_f32=False, skip=link), residual=x, skip=link)


class LinearBlock(nn.Module):
    """networks.py:523-568 (norm 'none' only)."""

    def __init__(self, input_dim, output_dim, norm='none', activation='relu'):
        super().__init__()
        self.fc = nn.Linear(input_dim, output_dim, bias=True)
        if norm != 'none':
            _unsupported("LinearBlock norm %r" % norm)
        self.norm = None
        if activation not in ('relu', 'lrelu', 'tanh', 'none'):
            _unsupported("activation %r" % activation)
        self.activation_type = activation

    def forward(self, x):
        return ops.linear(x, self.fc.weight, self.fc.bias, act=self.activation_type)


##################################################################################
# Sequential Models
##################################################################################
class ResBlocks(nn.Module):
    def __init__(self, num_blocks, dim, norm='in', activation='relu', pad_type='zero'):
        super().__init__()
        self.model = nn.Sequential(*[ResBlock(dim, norm=norm, activation=activation, pad_type=pad_type)
                                     for _ in range(num_blocks)])

    def forward(self, x):
        for blk in self.model:
            x = blk(x)
        return x


class MLP(nn.Module):
    """networks.py:432-443."""

    def __init__(self, input_dim, output_dim, dim, n_blk, norm='none', activ='relu'):
        super().__init__()
        model = [LinearBlock(input_dim, dim, norm=norm, activation=activ)]
        for _ in range(n_blk - 2):
            model += [LinearBlock(dim, dim, norm=norm, activation=activ)]
        model += [LinearBlock(dim, output_dim, norm='none', activation='none')]
        self.model = nn.Sequential(*model)

    def forward(self, x):
        h = x.reshape(x.size(0), -1)
        for blk in self.model:
            h = blk(h)
        return h


##################################################################################
# Encoder and Decoders
##################################################################################
class StyleEncoder(nn.Module):
    """networks.py:337-353."""

    def __init__(self, n_downsample, input_dim, dim, style_dim, norm, activ, pad_type):
        super().__init__()
        model = [Conv2dBlock(input_dim, dim, 7, 1, 3, norm=norm, activation=activ, pad_type=pad_type)]
        for _ in range(2):
            model += [Conv2dBlock(dim, 2 * dim, 4, 2, 1, norm=norm, activation=activ, pad_type=pad_type)]
            dim *= 2
        for _ in range(n_downsample - 2):
            model += [Conv2dBlock(dim, dim, 4, 2, 1, norm=norm, activation=activ, pad_type=pad_type)]
        model += [nn.AdaptiveAvgPool2d(1)]   # placeholder keeping the Sequential index; ops.global_avgpool runs
        model += [nn.Conv2d(dim, style_dim, 1, 1, 0)]
        self.model = nn.Sequential(*model)
        self.output_dim = dim

    def forward(self, x):
        n = len(self.model)
        for i in range(n - 2):
            x = self.model[i](x)
        x = ops.global_avgpool(x)
        last = self.model[n - 1]
        return ops.conv2d(x, last.weight, last.bias, 1, 0, 'none')


class ContentEncoder(nn.Module):
    """networks.py:355-369."""

    def __init__(self, n_downsample, n_res, input_dim, dim, norm, activ, pad_type):
        super().__init__()
        model = [Conv2dBlock(input_dim, dim, 7, 1, 3, norm=norm, activation=activ, pad_type=pad_type)]
        for _ in range(n_downsample):
            model += [Conv2dBlock(dim, 2 * dim, 4, 2, 1, norm=norm, activation=activ, pad_type=pad_type)]
            dim *= 2
        model += [ResBlocks(n_res, dim, norm=norm, activation=activ, pad_type=pad_type)]
        self.model = nn.Sequential(*model)
        self.output_dim = dim

    def forward(self, x):
        mods = list(self.model)
        for i, m in enumerate(mods):
            if isinstance(m, Conv2dBlock):      # consumed by the next convolution only (the last one also feeds a skip connection)
                x = m(x, want_f32=not (i + 1 < len(mods) and isinstance(mods[i + 1], Conv2dBlock)))
            else:
                x = m(x)
        return x


class Decoder_V2_atten(nn.Module):
    """networks.py:374-415: AdaIN ResBlocks, 2x (upsample, conv-AdaIN-ReLU, conv-AdaIN-ReLU), three 1x1
    convs, mask/blend head."""

    def __init__(self, n_upsample, n_res, dim, output_dim, res_norm='adain', activ='relu', pad_type='zero',
                 num_of_mask_dim_to_add=1):
        super().__init__()
        self.num_of_mask_dim_to_add = num_of_mask_dim_to_add
        self.output_dim = output_dim
        self.n_upsample = n_upsample
        self.mask_s = []
        self.split_active = False
        model = [ResBlocks(n_res, dim, res_norm, activ, pad_type=pad_type)]
        for _ in range(n_upsample):
            model += [nn.Upsample(scale_factor=2)]      # placeholder: fused into the next conv's gather
            model += [Conv2dBlock(dim, dim // 2, 3, 1, 1, norm='adain', activation=activ, pad_type=pad_type)]
            dim //= 2
            model += [Conv2dBlock(dim, dim, 3, 1, 1, norm='adain', activation=activ, pad_type=pad_type)]
        model += [Conv2dBlock(dim, dim, 1, 1, 0, norm='none', activation=activ, pad_type=pad_type)]
        model += [Conv2dBlock(dim, dim, 1, 1, 0, norm='none', activation=activ, pad_type=pad_type)]
        model += [Conv2dBlock(dim, output_dim * num_of_mask_dim_to_add + num_of_mask_dim_to_add, 1, 1, 0,
                              norm='none', activation='tanh', pad_type=pad_type)]
        self.model = nn.Sequential(*model)

    def _split_blocks(self):
        blocks = []
        for blk in self.model[0].model:
            blocks += [blk.model[0], blk.model[1]]
        i = 1
        for _ in range(self.n_upsample):
            blocks += [self.model[i + 1], self.model[i + 2]]
            i += 3
        return blocks

    def prepare_split_weights(self):
        """Fill the per-weight-version caches the tape-free trunk reads (the summed-tap weights of the upsampling layers,
        ops.SplitWeights.upconv_weights) on the CURRENT stream -- Council_Trainer calls it before its side streams fork."""
        i = 1
        for _ in range(self.n_upsample):
            blk = self.model[i + 1]
            wmgr = getattr(blk, '_cg_wmgr', None)
            w = blk.conv.weight
            if ops.upconv_weight_ok(w, wmgr):
                wmgr.upconv_weights(w, ops._grp(w), ops.group_n())
            i += 3

    def _trunk_split(self, x, want_f32=True):
        """ResBlocks + upsampling convs on the split-precision path (tape-free passes only):
        every 3x3 conv reads {hi, lo} fp16 planes written by the AdaIN apply before it."""
        xs = ops.split_f16(x)
        for blk in self.model[0].model:
            _, zs = conv_block_split(blk.model[0], xs)
            x, xs = conv_block_split(blk.model[1], zs, residual=x, want_f32=True)
        i = 1
        for u in range(self.n_upsample):
            _, zs = conv_block_split(self.model[i + 1], xs, upsample=True)
            last = u + 1 == self.n_upsample
            x, xs = conv_block_split(self.model[i + 2], zs, want_f32=last and want_f32)
            i += 3
        return x, xs, i

    def forward(self, x, im_in, return_mask=False):
        # split-precision trunk: only inside Council_Trainer's tape-free passes, which keep the split weights current
        if self.split_active and not torch.is_grad_enabled() and all(_split_ok(b) for b in self._split_blocks()):
            # tape-free pass: split-precision trunk, then the 1x1 head + mask / blend as ONE kernel reading the trunk's
            # {hi, lo} planes (ops.decoder_head_x3; the layer-by-layer head below when the fused kernel does not take the shape)
            head = [self.model[k].conv for k in range(len(self.model) - 3, len(self.model))]
            wmgr = getattr(self.model[len(self.model) - 1], '_cg_wmgr', None)
            fused_ok = ops.FUSED_HEAD and wmgr is not None and self.output_dim == 3 and self.num_of_mask_dim_to_add == 3 \
                and head[0].in_channels == 64 and all(self.model[k].norm is None for k in range(len(self.model) - 3, len(self.model))) \
                and [self.model[k].activation_type for k in range(len(self.model) - 3, len(self.model))] == ['relu', 'relu', 'tanh']      # what head_fwd_x3_kernel computes
            y, ys, i = self._trunk_split(x, want_f32=not fused_ok)
            if fused_ok:
                out = ops.decoder_head_x3(ys, head, wmgr, im_in, self.output_dim, self.num_of_mask_dim_to_add)
                if out is None:
                    raise hip.HipError("fused decoder head refused a shape it was chosen for")
                new_im, self.mask_s = out
                if return_mask:
                    return new_im, self.mask_s
                return new_im
        else:
            y = self.model[0](x)
            i = 1
            for _ in range(self.n_upsample):          # each output is read by the next convolution only
                y = self.model[i + 1](y, upsample=True, want_f32=False)
                y = self.model[i + 2](y, want_f32=False)
                i += 3
        y = self.model[i](y)
        y = self.model[i + 1](y)
        new_x = self.model[i + 2](y)
        new_im, self.mask_s = ops.mask_blend(new_x, im_in, self.output_dim, self.num_of_mask_dim_to_add)
        if return_mask:
            if self.mask_s.shape[1] != 3:            # networks.py:410-412 (host-side display helper)
                k = self.mask_s.shape[1]
                self.mask_s = (torch.sum(self.mask_s, 1).unsqueeze(1).repeat(1, 3, 1, 1) / k)
            return new_im, self.mask_s
        return new_im


##################################################################################
# Generator
##################################################################################
class AdaINGen(nn.Module):
    """networks.py:223-330."""

    def __init__(self, input_dim, params, cuda_device='cuda:0'):
        super().__init__()
        dim = params['dim']
        style_dim = params['style_dim']
        self.n_downsample = params['n_downsample']
        n_res = params['n_res']
        self.activ = params['activ']
        pad_type = params['pad_type']
        mlp_dim = params['mlp_dim']
        self.do_my_style = params['do_my_style']
        self.cuda_device = cuda_device
        if self.do_my_style:
            _unsupported("gen.do_my_style")
        self.enc_style = StyleEncoder(4, input_dim, dim, style_dim, norm='none', activ=self.activ, pad_type=pad_type)
        self.enc_content = ContentEncoder(self.n_downsample, n_res, input_dim, dim, 'in', self.activ,
                                          pad_type=pad_type)
        self.dec = Decoder_V2_atten(self.n_downsample, n_res, self.enc_content.output_dim, input_dim,
                                    res_norm='adain', activ=self.activ, pad_type=pad_type,
                                    num_of_mask_dim_to_add=params['num_of_mask_dim_to_add'])
        self.mlp = MLP(input_dim=style_dim, output_dim=self.get_num_adain_params(self.dec), dim=mlp_dim, n_blk=3,
                       norm='none', activ=self.activ)
        # column ranges of the MLP output per AdaIN layer, in modules() order (networks.py:303-312):
        # [start, start+C) -> bias ("mean"), [start+C, start+2C) -> weight ("std")
        start = 0
        for m in self.dec.modules():
            if m.__class__.__name__ == "AdaptiveInstanceNorm2d":
                m.boff, m.goff = start, start + m.num_features
                start += 2 * m.num_features

    def forward(self, images, style, return_mask=False):
        content, _ = self.encode(images)
        return self.decode(content, style, images, return_mask=return_mask)

    def encode_content(self, images):
        """Content code only.  The hot path never uses the style code (the
        StyleEncoder output is discarded when recon_s_w = recon_x_w = 0), so the trainer calls this."""
        return self.enc_content(images)

    def encode(self, images):
        # networks.py:278-283
        style_fake = self.enc_style(images)
        content = self.enc_content(images)
        return content, style_fake

    def decode(self, content, style, images, return_mask=False):
        # networks.py:285-301
        adain_params = self.mlp(style)
        self.assign_adain_params(adain_params, self.dec)
        return self.dec(content, images, return_mask)

    def assign_adain_params(self, adain_params, model):
        mods = self.__dict__.get('_adain_mods')
        if mods is None or mods[0] is not model:
            mods = (model, [m for m in model.modules() if m.__class__.__name__ == "AdaptiveInstanceNorm2d"])
            self.__dict__['_adain_mods'] = mods
        # one gradient buffer for the whole matrix instead of a full-size gradient per layer added up by the engine (ops.ParamGrad)
        adain_params, pgrad = ops.adain_param_fork(adain_params)
        for m in mods[1]:
            d = m.__dict__                     # plain attributes: skip nn.Module.__setattr__'s type dispatch
            d['params'] = adain_params
            d['pgrad'] = pgrad
            d['bias'] = adain_params[:, m.boff:m.boff + m.num_features]        # views, no copy
            d['weight'] = adain_params[:, m.goff:m.goff + m.num_features]

    def get_num_adain_params(self, model):
        return sum(2 * m.num_features for m in model.modules() if m.__class__.__name__ == "AdaptiveInstanceNorm2d")


##################################################################################
# Discriminators
##################################################################################
class _LossVectors:
    """Per-sample LSGAN target / weight vectors on the device, cached by value."""

    def __init__(self):
        self._cache = {}

    def get(self, tgt, wt, device):
        key = (tuple(tgt), tuple(wt), str(device))
        v = self._cache.get(key)
        if v is None:
            # published only once the copies have completed: the module is called from more than one stream (hip.upload_const)
            v = (hip.upload_const(torch.tensor(tgt, dtype=torch.float32)), hip.upload_const(torch.tensor(wt, dtype=torch.float32)))
            hip.const_cache_put(self._cache, key, v)
        return v


class MsImageDis(nn.Module):
    """Multi-scale PatchGAN discriminator, networks.py:17-110 (LSGAN)."""

    def __init__(self, input_dim, params, cuda_device='cuda:0'):
        super().__init__()
        self.n_layer = params['n_layer']
        self.gan_type = params['gan_type']
        self.dim = params['dim']
        self.norm = params['norm']
        self.activ = params['activ']
        self.num_scales = params['num_scales']
        self.pad_type = params['pad_type']
        self.cuda_device = cuda_device
        self.input_dim = input_dim
        if self.gan_type != 'lsgan':
            _unsupported("gan_type %r" % self.gan_type)
        self.downsample = nn.AvgPool2d(3, stride=2, padding=[1, 1], count_include_pad=False)  # placeholder
        self.cnns = nn.ModuleList()
        for _ in range(self.num_scales):
            self.cnns.append(self._make_net())
        self._vec = _LossVectors()

    def _make_net(self):
        dim = self.dim
        cnn_x = [Conv2dBlock(self.input_dim, dim, 4, 2, 1, norm='none', activation=self.activ, pad_type=self.pad_type)]
        for _ in range(self.n_layer - 1):
            cnn_x += [Conv2dBlock(dim, dim * 2, 4, 2, 1, norm=self.norm, activation=self.activ, pad_type=self.pad_type)]
            dim *= 2
        cnn_x += [nn.Conv2d(dim, 1, 1, 1, 0)]
        return nn.Sequential(*cnn_x)

    def forward(self, x):
        outputs = []
        for si, model in enumerate(self.cnns):
            y = x
            for blk in list(model)[:-1]:
                y = blk(y, want_f32=False)           # every block's output is read by the next convolution only
            last = model[len(model) - 1]
            outputs.append(ops.conv2d(y, last.weight, last.bias, 1, 0, 'none', wmgr=getattr(self, '_cg_wmgr', None)))
            if si + 1 < len(self.cnns):
                x = ops.avgpool3s2(x)
        return outputs

    def calc_dis_loss(self, input_fake, input_real, weight=1.0):
        """networks.py:56-82.  Fake and real run as ONE batch (the net has no cross-sample op).  Under ops.members(n)
        `input_fake` holds the n members' fakes (member-major), `input_real` the one real batch they all see; the
        discriminator batch is [fake_0 | real | fake_1 | real | ...] and the result one loss per member."""
        n = ops.group_n()
        b, br = input_fake.shape[0] // n, input_real.shape[0]
        idx = []
        for m in range(n):
            idx += list(range(m * b, (m + 1) * b)) + [-r - 1 for r in range(br)]
        outs = self.forward(ops.take_rows(input_fake, input_real, idx))
        tgt, wt = self._vec.get(([0.0] * b + [1.0] * br) * n, [weight] * (n * (b + br)), input_fake.device)
        return ops.lsgan_loss(outs, tgt, wt, b)

    def calc_gen_loss(self, input_fake, input_real=None, weight=1.0):
        """networks.py:84-110 (one loss per member under ops.members(n))."""
        nb = input_fake.shape[0]
        outs = self.forward(input_fake)
        tgt, wt = self._vec.get([1.0] * nb, [weight] * nb, input_fake.device)
        return ops.lsgan_loss(outs, tgt, wt, nb // ops.group_n())


class MsImageDisCouncil(nn.Module):
    """Council (conditional) discriminator, networks.py:116-215: the candidate image and the input image
    enter the first 3x3 conv as a two-source gather instead of a materialised 6-channel concat."""

    def __init__(self, input_dim, params, cuda_device='cuda:0'):
        super().__init__()
        self.n_layer = params['n_layer']
        self.gan_type = params['gan_type']
        self.dim = params['dim']
        self.norm = params['norm']
        self.activ = params['activ']
        self.num_scales = params['num_scales']
        self.pad_type = params['pad_type']
        self.cuda_device = cuda_device
        self.input_dim = input_dim
        if self.gan_type != 'lsgan':
            _unsupported("gan_type %r" % self.gan_type)
        self.downsample = nn.AvgPool2d(3, stride=2, padding=[1, 1], count_include_pad=False)  # placeholder
        self.cnns = nn.ModuleList()
        for _ in range(self.num_scales):
            self.cnns.append(self._make_net())
        self._vec = _LossVectors()

    def _make_net(self):
        dim = self.dim
        cnn_x = [Conv2dBlock(2 * self.input_dim, dim, 3, 1, 1, norm='none', activation=self.activ,
                             pad_type=self.pad_type)]
        for _ in range(self.n_layer - 1):
            cnn_x += [Conv2dBlock(dim, dim * 2, 4, 2, 1, norm=self.norm, activation=self.activ, pad_type=self.pad_type)]
            dim *= 2
        cnn_x += [nn.Conv2d(dim, dim, 1, 1, 0)]
        cnn_x += [nn.Conv2d(dim, 1, 1, 1, 0)]
        return nn.Sequential(*cnn_x)

    def forward(self, x, x_input):
        outputs = []
        for si, model in enumerate(self.cnns):
            blocks = list(model)
            tail = ops.composed_tail_ok(blocks[-2], blocks[-1])
            y = blocks[0](x, x2=x_input, want_f32=False)      # every block's output is read by the next convolution only ...
            for k, blk in enumerate(blocks[1:-2]):
                y = blk(y, want_f32=tail and k == len(blocks) - 4)      # ... except in front of the composed fp32 tail
            wmgr = getattr(self, '_cg_wmgr', None)
            if tail:
                # Conv2d(dim, dim, 1) -> Conv2d(dim, 1, 1) with no activation in between (networks.py:142-143) is ONE dim -> 1
                # convolution with the product of the two weight matrices: the dim -> dim layer is never run
                outputs.append(ops.composed_tail(y, blocks[-2], blocks[-1]))
            else:
                y = ops.conv2d(y, blocks[-2].weight, blocks[-2].bias, 1, 0, 'none', wmgr=wmgr, want_split=wmgr is not None)
                outputs.append(ops.conv2d(y, blocks[-1].weight, blocks[-1].bias, 1, 0, 'none', wmgr=wmgr))
            if si + 1 < len(self.cnns):
                x = ops.avgpool3s2(x)
                x_input = ops.avgpool3s2(x_input)
        return outputs

    def calc_dis_loss(self, input_fake, input_real, input, weight=1.0):
        """networks.py:158-186 (reference signature)."""
        return self.calc_dis_loss_multi(input_fake, [input_real], [1.0], input, fake_weight=1.0, weight=weight)

    def calc_dis_loss_multi(self, input_fake, reals, real_weights, input, fake_weight, weight=1.0):
        """sum_k [ mean(D(fake)^2) + mean((D(real_k) - 1)^2) ] with repeated terms folded into weights:
        the reference re-runs the identical fake pass for every colleague pick (trainer_council.py:
        862-874); here fake and the distinct colleagues' images run once, as one batch."""
        b = input_fake.shape[0]
        pool = torch.cat(list(reals), 0) if len(reals) > 1 else (reals[0] if reals else None)     # API convenience only:
        picks = [[(k, rw) for k, rw in enumerate(real_weights)]]         # the trainer calls calc_dis_loss_members directly
        return self.calc_dis_loss_members(input_fake, pool, picks, input, fake_weight, weight)

    def calc_dis_loss_members(self, x_full, x_cmp, picks, input, fake_weight, weight=1.0):
        """The council-discriminator objective of n members as one batch (n = ops.group_n() = len(picks)).
        x_full: the members' own translations, member-major [n*B]; x_cmp: comparison images, block j = rows [j*B, (j+1)*B);
        picks[m]: [(block j, multiplicity), ...] -- the DISTINCT colleagues member m drew (trainer_council.py:861-868)
        with how often; input: the one conditioning batch [B].  Discriminator batch of member m:
        [own | cmp_j1 | cmp_j2 | ...] with per-sample loss weights; every member has the same number of blocks."""
        n = ops.group_n()
        b = x_full.shape[0] // n
        idx, idx_in, tgt, wt = self.plan_members(picks, n, b, fake_weight, weight)
        tgt, wt = self._vec.get(tgt, wt, x_full.device)
        return self.calc_dis_loss_planned(x_full, x_cmp, input, idx, idx_in, tgt, wt, None)

    @staticmethod
    def plan_members(picks, n, b, fake_weight, weight=1.0):
        """Host half of calc_dis_loss_members: (rows of the discriminator batch as take_rows indices into [x_full | x_cmp],
        rows of the conditioning batch, LSGAN targets, per-sample loss weights) as plain lists."""
        if len(picks) != n:
            raise ValueError("one pick list per member of the launch")
        u = len(picks[0])
        if any(len(p) != u for p in picks):
            raise ValueError("members of one launch must compare against the same number of distinct colleagues")
        idx, idx_in, tgt, wt = [], [], [], []
        for m in range(n):
            idx += list(range(m * b, (m + 1) * b))
            tgt += [0.0] * b
            wt += [weight * fake_weight] * b
            for j, mult in picks[m]:
                idx += [-(j * b + r) - 1 for r in range(b)]
                tgt += [1.0] * b
                wt += [weight * mult] * b
            idx_in += list(range(b)) * (1 + u)
        return idx, idx_in, tgt, wt

    def calc_dis_loss_planned(self, x_full, x_cmp, input, idx, idx_in, tgt_dev, wt_dev, idx_dev):
        """Device half: `idx` / `idx_in` from plan_members (`idx_dev`: the same rows as a device vector the caller keeps
        current -- the colleague picks change every iteration, graphs.HostInputs), targets / weights as device vectors."""
        b = x_full.shape[0] // ops.group_n()
        x = ops.take_rows(x_full, x_cmp, idx, idx_dev=idx_dev)
        x_in = ops.take_rows(input, None, idx_in)
        outs = self.forward(x, x_in)
        return ops.lsgan_loss(outs, tgt_dev, wt_dev, b)

    def calc_gen_loss(self, input_fake, input, input_real=None, weight=1.0):
        """networks.py:188-215 (one loss per member under ops.members(n); `input` is then the member-major repetition
        of the conditioning batch)."""
        nb = input_fake.shape[0]
        outs = self.forward(input_fake, input)
        tgt, wt = self._vec.get([1.0] * nb, [weight] * nb, input_fake.device)
        return ops.lsgan_loss(outs, tgt, wt, nb // ops.group_n())
