"""RCNetModel on MI355X: same constructor, methods and checkpoint keys as the reference's RCNet/rcnet_model.py
(RCNetModel :6, forward :101, compute_loss :125, save_model :211, restore_model :234, data_parallel :259).
`log_summary` (TensorBoard images) is out of scope.
"""
import torch

from . import engine, networks


class RCNetModel(object):
    def __init__(self, input_channels_image, input_channels_depth, input_patch_size_image, encoder_type,
                 n_filters_encoder_image, n_neurons_encoder_depth, decoder_type, n_filters_decoder,
                 weight_initializer='kaiming_uniform', activation_func='leaky_relu', device=torch.device('cuda')):
        self.input_patch_size_image = input_patch_size_image
        self.device = device
        height, width = input_patch_size_image
        latent_height = int((height // 32.0))
        latent_width = int((width // 32.0))
        latent_size_depth = latent_height * latent_width * n_neurons_encoder_depth[-1]

        if 'rcnet' in encoder_type:
            self.encoder = networks.RCNetEncoder(
                input_channels_image=input_channels_image, input_channels_depth=input_channels_depth,
                input_patch_size_image=input_patch_size_image, n_filters_encoder_image=n_filters_encoder_image,
                n_neurons_encoder_depth=n_neurons_encoder_depth, latent_size_depth=latent_size_depth,
                weight_initializer=weight_initializer, activation_func=activation_func,
                use_batch_norm='batch_norm' in encoder_type)
        else:
            raise ValueError('Encoder type {} not supported.'.format(encoder_type))

        n_skips = n_filters_encoder_image[:-1]
        n_skips = n_skips[::-1] + [0]
        latent_channels = n_filters_encoder_image[-1] + n_neurons_encoder_depth[-1]

        if 'multiscale' in decoder_type:
            self.decoder = networks.MultiScaleDecoder(
                input_channels=latent_channels, output_channels=1, n_resolution=1, n_filters=n_filters_decoder,
                n_skips=n_skips, weight_initializer=weight_initializer, activation_func=activation_func,
                output_func='linear', use_batch_norm='batch_norm' in decoder_type, deconv_type='up')
        else:
            raise ValueError('Decoder type {} not supported.'.format(decoder_type))
        self._ddp = None
        self.to(self.device)

    # -- reference: rcnet_model.py:101-123 --------------------------------------------------------------
    def forward(self, image, point, bounding_boxes, return_logits=True):
        """image N x 3 x H x W, point (N*K) x 3, bounding_boxes list of N (K,4) tensors -> (N*K) x 1 x h x w logits.
        Encoder and decoder run as ONE region so pooled skips never leave the activation dtype."""
        enc, dec = self._unwrap(self.encoder), self._unwrap(self.decoder)
        rois = networks.boxes_to_rois(bounding_boxes)
        shape = self.input_patch_size_image

        def run(image, point, rois):
            pts = point if point.is_contiguous() else point.contiguous()
            latent, skips = enc._fwd(engine.from_nchw(image), pts, rois)
            engine.stage_mark("decoder_done")     # backward: every decoder gradient is final here (parallel.GradientAllReducer)
            logits = dec._fwd(latent, skips, shape)[-1]
            return engine.to_nchw_out(logits, torch.float32)
        params = list(enc.parameters()) + list(dec.parameters())
        # (the RoI rows are a region INPUT: a captured region -- engine.set_autograph -- reads them from a static tensor)
        logits = engine.run_region(run, (image, point, rois), params, graph_key=("RCNetModel.forward", id(enc), id(dec), enc.training, dec.training, tuple(shape)),
                                   on_replay=self._bn_replay)
        if return_logits:
            return logits
        lg = logits.contiguous()
        return engine.sigmoid(lg)

    # -- reference: rcnet_model.py:125-166 ---------------------------------------------------------------
    def compute_loss(self, logits, ground_truth, validity_map, w_positive_class=1.0):
        gt = ground_truth if ground_truth.is_contiguous() else ground_truth.contiguous()
        vm = validity_map if validity_map.is_contiguous() else validity_map.contiguous()

        def run(logits):
            return engine.bce_masked(logits, gt, vm, float(w_positive_class))
        loss = engine.run_region(run, (logits,), [])
        return loss, {'loss': loss}

    @staticmethod
    def _unwrap(m):
        return m.module if hasattr(m, 'module') else m

    def _bn_replay(self):
        """host-side bookkeeping of one replayed forward: the BatchNorm layers' num_batches_tracked counters (net_utils._BNCounter)"""
        for net in (self._unwrap(self.encoder), self._unwrap(self.decoder)):
            for m in net.modules():
                if getattr(m, 'use_batch_norm', False) and hasattr(m, '_nbt_pending') and m.training:
                    m._nbt_pending += 1

    def parameters(self):
        return list(self.encoder.parameters()) + list(self.decoder.parameters())

    def train(self):
        self.encoder.train()
        self.decoder.train()

    def eval(self):
        self.encoder.eval()
        self.decoder.eval()

    def to(self, device):
        self.encoder.to(device)
        self.decoder.to(device)

    # -- reference: rcnet_model.py:211-257 (same checkpoint dict keys) --------------------------------------
    def save_model(self, checkpoint_path, step, optimizer):
        engine.check_roi_overflow()      # the host waits here anyway: a geometry the one-byte RoI arg-max cannot encode raises instead of saving NaNs
        checkpoint = {}
        checkpoint['train_step'] = step
        checkpoint['radarnet_optimizer_state_dict'] = optimizer.state_dict()
        checkpoint['radarnet_encoder_state_dict'] = self.encoder.state_dict()
        checkpoint['radarnet_decoder_state_dict'] = self.decoder.state_dict()
        torch.save(checkpoint, checkpoint_path)

    def restore_model(self, checkpoint_path, optimizer=None):
        checkpoint = torch.load(checkpoint_path, map_location=self.device)
        self.encoder.load_state_dict(checkpoint['radarnet_encoder_state_dict'])
        self.decoder.load_state_dict(checkpoint['radarnet_decoder_state_dict'])
        if optimizer is not None:
            optimizer.load_state_dict(checkpoint['radarnet_optimizer_state_dict'])
        engine.refresh_packed()   # cached MFMA operands follow the restored weights in place (captured hipGraphs keep their addresses)
        return checkpoint['train_step'], optimizer

    # -- reference: rcnet_model.py:259-265 -------------------------------------------------------------------
    def data_parallel(self):
        """The reference wraps encoder/decoder in torch.nn.DataParallel (single process).  Here data parallelism is
        one process per GPU: gradients are all-reduced over RCCL by riders_amd.parallel.GradientAllReducer.  The
        wrappers keep the `module.` prefix the reference's checkpoints carry (rcnet_main.py:145,423)."""
        from .parallel import ModuleHolder
        self.encoder = ModuleHolder(self.encoder)
        self.decoder = ModuleHolder(self.decoder)
