"""``task: bg_forecast`` — the two reference tasks fused on the device (new; no reference counterpart).

In the reference the hot path is split by an on-disk hop: ``task: pc_transform`` runs once per input
frame (``only_this_ind`` 0/1/2, configs/bg/bg_val_short.yaml:12-14) and exports label PNGs + u16 depth
PNGs (export_cityscapes_segmentation_results.py:108-124); ``task: bg`` reads them back through
BGDataset (bg_dataset.py:203-230).  Here one ``predict`` does

    3 x warp/splat (per-frame z-buffers, one launch)  ->  [hop emulated in registers]  ->  HarDNet -> argmax

with nothing but the final label map leaving the GPU path.  ``model.emulate_disk_hop`` (default True)
reproduces the quantisation the PNG/H5 round trip applies (id->trainId LUT, depth -> u16 -> /256-1,
mask = d>0, clamp to [min_depth, max_depth]) so outputs match the two-stage reference pipeline.
"""
import torch

from .model_api import BaseModel
from .bg_model import BGModel, LazyResult
from .pc_transform_model import WarpSplat

PF_HOP_TRAINID_LUT = 1
PF_HOP_DEPTH_U16 = 2


class BGForecastModel(BaseModel):

    def __init__(self, params):
        super().__init__()
        mp = params['model']
        self.bg = BGModel(params)
        self.emulate_disk_hop = mp.get('emulate_disk_hop', True)
        self.seg_is_label_id = mp.get('seg_is_label_id', True)   # export run used --convert_to_trainid
        self.return_logits = mp.get('return_logits', False)   # True: full-size + network-size logits; 'orig': network-size only
        self.per_sample_sentinel = bool(mp.get('per_sample_sentinel', False))   # see pc_transform_model.PCTransformModel
        self._splat = WarpSplat()
        # optional callable run between the warp/splat launches and the network launches of a predict() - a caller that
        # pipelines several sub-batches on several streams records its cross-stream event here (bench.py --stagger)
        self.after_splat = None

    # checkpoint compatibility: a reference bg_model.pt loads straight into the fused model
    def load_state_dict(self, state_dict, strict=True):
        if not any(k.startswith('bg.') for k in state_dict):
            return self.bg.load_state_dict(state_dict, strict)
        return super().load_state_dict(state_dict, strict)

    @torch.no_grad()
    def predict(self, inputs, labels=None):
        seg_w, depth_w, _ = self._splat(inputs['depth'], inputs['depth_mask'], inputs['seg'],
                                        inputs['intrinsics'], inputs['extrinsics'], inputs['target_T'],
                                        Kinv=inputs.get('intrinsics_inv'), Einv=inputs.get('extrinsics_inv'),
                                        per_frame=True, want_result2d=False,
                                        per_sample_sentinel=self.per_sample_sentinel)
        if self.after_splat is not None:
            self.after_splat()
        hop = 0
        if self.emulate_disk_hop:
            hop |= PF_HOP_DEPTH_U16
        if self.seg_is_label_id:
            hop |= PF_HOP_TRAINID_LUT
        mask = None if (hop & PF_HOP_DEPTH_U16) else (depth_w > 0)
        (seg, logits, orig), token = self.bg.run_async(seg_w, depth_w, mask, want_logits=self.return_logits is True,
                                                       want_orig=bool(self.return_logits), hop_flags=hop, seg_dtype=torch.uint8,
                                                       own_inputs=True)
        out = {'seg': seg, 'warped_seg': seg_w, 'warped_depth': depth_w}
        if logits is not None:
            out['logits'] = logits
        if orig is not None:
            out['orig_size_logits'] = orig
        return LazyResult(out, self.bg, token)     # nothing waited for: the range check happens on first access
