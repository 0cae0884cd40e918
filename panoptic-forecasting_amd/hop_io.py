"""On-disk formats either side of the hop between ``task: pc_transform`` and ``task: bg``.

The reference's two bg stages talk through files:

  export side  experiments/export_cityscapes_segmentation_results.py:75-127 (``export_results``): per target frame
               ``{city}_{seq}_{frame:06d}_gtFine_labelIds.png`` (u8 label map; trainId -> id unless ``--no_convert``,
               id -> trainId with ``--convert_to_trainid``), ``..._leftImg8bit.png`` (``--is_img``) and, with
               ``--save_depth_as_png``, ``..._depths.png`` = uint16 ``round(clamp(d+1, 0, 255) * 256)``
               (``--save_depth`` alone: ``..._depths.npy`` float32);
  load side    data/datasets/bg_dataset.py:172-232 (``BGDataset.__getitem__``): T label PNGs -> ``seg [T,H,W]`` int64,
               depth from an H5 dataset ``{city}/{seq}/{frame:06d}/{start_frame}`` of shape ``[H,W,T]`` (same u16
               code) -> ``depth = x/256 - 1``, ``depth_mask = depth > 0``, ``depth[~mask] = -1``, clamp to
               ``[min_depth, max_depth]``.

Here the label conversion, the u16 quantisation and the decode run on the device (``pf_hop_export`` /
``pf_hop_load``, csrc/hop_kernels.hip): 3 B per pixel cross PCIe on export instead of 5.  PNG (de)compression is
PIL's, as in the reference (cv2 for the u16 files there; both write plain 16-bit greyscale PNGs).  H5 access needs
``h5py``, which this image lacks: ``DepthH5`` raises a clear error when it is missing.
"""
import os

import numpy as np
import torch

from . import lib as _lib

LABEL_PNG = '%s_%s_%06d_gtFine_labelIds.png'
IMAGE_PNG = '%s_%s_%06d_leftImg8bit.png'
DEPTH_PNG = '%s_%s_%06d_depths.png'
DEPTH_NPY = '%s_%s_%06d_depths.npy'

SEG_AS_IS, SEG_TRAINID_TO_ID, SEG_ID_TO_TRAINID = 0, 1, 2


def seg_mode(no_convert=False, convert_to_trainid=False, is_img=False):
    """The branch structure of export_results :91-94."""
    if is_img:
        return SEG_AS_IS
    if not no_convert:
        return SEG_TRAINID_TO_ID
    return SEG_ID_TO_TRAINID if convert_to_trainid else SEG_AS_IS


def h5_key(city, seq, frame, start_frame):
    """bg_dataset.py:118,186: ``'%s/%s/%06d/%d'`` — start_frame is a float there (``(9-gap_len)/3``) formatted with %d."""
    return '%s/%s/%06d/%d' % (city, seq, frame, start_frame)


# ---------------------------------------------------------------------------------------------- device side
def device_export(seg=None, depth=None, mode=SEG_AS_IS):
    """(seg u8 | None, depth u16-coded int16 | None) on the device, shapes preserved.  seg: u8 or i64; depth: f32."""
    L = _lib.load()
    n = (seg if seg is not None else depth).numel()
    out_seg = out_q = None
    seg_ptr = dep_ptr = None
    is64 = 0
    if seg is not None:
        if seg.dtype not in (torch.uint8, torch.int64):
            seg = seg.to(torch.int64)
        seg = _lib.require_cuda(seg.contiguous(), 'seg')
        is64 = int(seg.dtype == torch.int64)
        out_seg = torch.empty(seg.shape, dtype=torch.uint8, device=seg.device)
        seg_ptr = seg.data_ptr()
    if depth is not None:
        depth = _lib.require_cuda(depth.float().contiguous(), 'depth')
        out_q = torch.empty(depth.shape, dtype=torch.int16, device=depth.device)   # u16 bit patterns
        dep_ptr = depth.data_ptr()
    _lib.check(L.pf_hop_export(seg_ptr, is64, int(mode), dep_ptr, n,
                               out_seg.data_ptr() if out_seg is not None else None,
                               out_q.data_ptr() if out_q is not None else None, _lib.stream_ptr()), 'pf_hop_export')
    return out_seg, out_q


def device_load_depth(q, min_depth, max_depth):
    """q: u16 codes as an int16/uint16 device tensor -> (depth f32, depth_mask bool)."""
    L = _lib.load()
    q = _lib.require_cuda(q.contiguous(), 'depth codes')
    if q.element_size() != 2:
        raise _lib.PfError('depth codes must be 16-bit (got %s)' % q.dtype)
    depth = torch.empty(q.shape, dtype=torch.float32, device=q.device)
    mask = torch.empty(q.shape, dtype=torch.uint8, device=q.device)
    _lib.check(L.pf_hop_load(q.data_ptr(), q.numel(), float(min_depth), float(max_depth), depth.data_ptr(),
                             mask.data_ptr(), _lib.stream_ptr()), 'pf_hop_load')
    return depth, mask.view(torch.bool)


def u16_numpy(q):
    """int16-typed device/host tensor holding u16 codes -> numpy uint16."""
    return q.detach().cpu().numpy().view(np.uint16)


# ---------------------------------------------------------------------------------------------- files
def write_png(path, arr):
    from PIL import Image
    Image.fromarray(arr).save(path)      # uint8 [H,W] / [H,W,3] -> 8-bit, uint16 [H,W] -> 16-bit greyscale


def read_png(path):
    from PIL import Image
    return np.array(Image.open(path))


def export_batch(preds, meta, base_result_dir, no_convert=False, convert_to_trainid=False, is_img=False,
                 save_depth=False, save_depth_as_png=False):
    """The per-batch body of export_results (:86-127) for device-resident ``preds`` of PCTransformModel / BGModel.
    Returns the list of label/image files written."""
    mode = seg_mode(no_convert, convert_to_trainid, is_img)
    want_q = save_depth and save_depth_as_png
    seg8, q = device_export(preds['seg'], preds['depth'] if want_q else None, mode)
    seg_np = seg8.cpu().numpy()
    q_np = u16_numpy(q) if q is not None else None
    written = []
    for b in range(seg_np.shape[0]):
        city, seq, target = meta['city'][b], meta['seq'][b], int(meta['target_frame'][b])
        out_dir = os.path.join(base_result_dir, city)
        os.makedirs(out_dir, exist_ok=True)
        name = (IMAGE_PNG if is_img else LABEL_PNG) % (city, seq, target)
        write_png(os.path.join(out_dir, name), seg_np[b])
        written.append(os.path.join(out_dir, name))
        if save_depth:
            if save_depth_as_png:
                write_png(os.path.join(out_dir, DEPTH_PNG % (city, seq, target)), q_np[b])
            else:
                np.save(os.path.join(out_dir, DEPTH_NPY % (city, seq, target)), preds['depth'][b].cpu().numpy())
    return written


def fill_missing(base_result_dir, gt_split_dir, cities=None, background_dir=None, no_convert=False, shape=(1024, 2048)):
    """export_results :129-165: every ground-truth frame without a prediction gets the background file (converted
    trainId -> id) or a constant map (255 with no_convert, else 0).  Returns the number of files created."""
    import glob
    lut = np.zeros(256, np.uint8)
    lut[:19] = [7, 8, 11, 12, 13, 17, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 31, 32, 33]
    count = 0
    for city in sorted(os.listdir(gt_split_dir)):
        if cities is not None and city not in cities:
            continue
        for gt_path in sorted(glob.glob(os.path.join(gt_split_dir, city, '*_gtFine_labelIds.png'))):
            fname = os.path.basename(gt_path)
            out_name = os.path.join(base_result_dir, city, fname)
            if os.path.exists(out_name):
                continue
            count += 1
            src = os.path.join(background_dir, city, fname) if background_dir else None
            if src and os.path.exists(src):
                img = lut[read_png(src).astype(np.uint8)]
            else:
                img = np.full(shape, 255 if no_convert else 0, np.uint8)
            os.makedirs(os.path.dirname(out_name), exist_ok=True)
            write_png(out_name, img)
    return count


class DepthH5:
    """``[H,W,T]`` u16-coded depth stacks keyed ``city/seq/%06d/%d`` (bg_dataset.py:183-187)."""

    def __init__(self, path, mode='r'):
        try:
            import h5py
        except ImportError as e:   # this image has no h5py; the format is still specified above
            raise ImportError('DepthH5 needs h5py, which is not installed here; use depth PNG stacks '
                              '(load_bg_inputs(depth_pngs=...)) instead') from e
        self._f = h5py.File(path, mode)

    def read(self, city, seq, frame, start_frame):
        return self._f[h5_key(city, seq, frame, start_frame)][:]

    def write(self, city, seq, frame, start_frame, stack_hw_t):
        self._f.create_dataset(h5_key(city, seq, frame, start_frame), data=stack_hw_t.astype(np.uint16), compression='gzip')

    def close(self):
        self._f.close()


def load_bg_inputs(label_pngs, depth_pngs=None, depth_stack=None, min_depth=0.1, max_depth=200.0, device='cuda'):
    """BGDataset.__getitem__ (:172-232) for one sample: T label PNGs + the T depth codes (as T u16 PNGs or as the
    H5 ``[H,W,T]`` stack) -> ``{'seg' [T,H,W] i64, 'depth' [T,H,W] f32, 'depth_mask' [T,H,W] bool}`` on ``device``;
    the depth decode runs on the device."""
    seg = torch.from_numpy(np.stack([read_png(p) for p in label_pngs])).to(device).long()
    out = {'seg': seg}
    if depth_pngs is not None:
        depth_stack = np.stack([read_png(p) for p in depth_pngs], axis=2)
    if depth_stack is not None:
        codes = np.ascontiguousarray(np.moveaxis(np.asarray(depth_stack), 2, 0).astype(np.uint16))   # :199-201
        q = torch.from_numpy(codes.view(np.int16)).to(device)
        out['depth'], out['depth_mask'] = device_load_depth(q, min_depth, max_depth)
    return out


def collate(samples):
    """bg_dataset.py:235-261 for the tensors this module produces."""
    return {k: torch.stack([s[k] for s in samples]) for k in samples[0]}
