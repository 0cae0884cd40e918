"""Captured calls replay to the same bits as eager ones - on EVERY replay, not only the first.

Round 4 found that a captured hipMemsetAsync node inside a graph that is one linear chain (one stream, no forks) executed on
the first launch of the graph and not on later ones (torch 2.10 capture + replay; tools/ubench/graph_memset_torch.py): the splat's z-buffer slots and the training step's gradient
arenas held garbage from the second replay on, while the forked four-stream graph of the headline bench was not
affected.  The library now enqueues kernels only (csrc/pf_fill.hip); these tests replay each captured path five times."""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), 'golden')
REPLAYS = 5


def _forecast_model(h, w, **model_kw):
    from panoptic_forecasting_amd import synth
    from panoptic_forecasting_amd.registry import build_model
    with open(os.path.join(G, 'calib_seed1234.json')) as f:
        sd = synth.make_state_dict(seed=1234, calib=json.load(f))
    params = {'task': 'bg_forecast', 'no_gpu': False, 'load_model': None, 'load_best_model': False,
              'data': {'num_classes': 11, 'depth_norm_params': [torch.tensor([20.]), torch.tensor([15.])],
                       'min_depth': 0.1, 'max_depth': 200},
              'model': dict({'num_inputs': 3, 'use_depth_inps': True, 'convert2onehot': True, 'final_h': h, 'final_w': w}, **model_kw)}
    m = build_model(params)
    m.load_state_dict(sd)
    m.eval()
    return m


def _tensors(res):
    return {k: v.clone() for k, v in dict(res).items() if torch.is_tensor(v)}


@pytest.mark.parametrize('streams', [1, 2])
def test_forecast_graph_replays_equal_eager(streams):
    """bench.py's launch form: `streams` sub-batches, each through its own model on its own stream, one captured graph
    (streams = 1: a linear chain of nodes, the form of the by_batch legs)."""
    from panoptic_forecasting_amd import synth
    h, w, b = 256, 512, 2
    models = [_forecast_model(h, w, return_logits=True) for _ in range(streams)]
    subs = [{k: v.cuda() for k, v in synth.make_inputs(b=b, h=h, w=w, seed=20 + i, gap_len=3).items()} for i in range(streams)]
    side = [torch.cuda.Stream() for _ in range(streams - 1)]

    def step():
        cur = torch.cuda.current_stream()
        outs = [None] * streams
        for i in range(1, streams):
            side[i - 1].wait_stream(cur)
            with torch.cuda.stream(side[i - 1]):
                outs[i] = models[i].predict(subs[i], None)
        outs[0] = models[0].predict(subs[0], None)
        for i in range(1, streams):
            cur.wait_stream(side[i - 1])
        return outs

    want = [_tensors(o) for o in step()]
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    for m in models:
        m.bg.settle()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        outs = step()
    for rep in range(REPLAYS):
        for o in outs:                      # a replay that skipped its work would otherwise be read as the previous one
            for v in dict(o).values():
                if torch.is_tensor(v):
                    v.fill_(77)
        g.replay()
        torch.cuda.synchronize()
        for i, o in enumerate(outs):
            got = _tensors(o)
            assert set(got) == set(want[i])
            for k in got:
                assert torch.equal(got[k], want[i][k]), (rep, i, k, int((got[k] != want[i][k]).sum()))
    for m in models:
        assert m.bg.range_status_sticky() == 0


def test_panoptic_encode_graph_replays_equal_eager():
    """pf_panoptic_encode zeroes its `present` table at the start of every call: through the C ABI, captured."""
    from panoptic_forecasting_amd import lib as pflib
    L = pflib.load()
    gen = torch.Generator().manual_seed(3)
    b, h, w = 2, 64, 128
    seg = torch.randint(0, 11, (b, h, w), generator=gen).cuda()
    seg[0, :20, :30] = 11003
    seg[1, 40:, 100:] = 17001
    max_ids = L.pf_panoptic_max_ids()

    def buffers():
        return (torch.full((b, h, w, 3), 9, dtype=torch.uint8, device='cuda'), torch.full((b, h, w), 9, dtype=torch.int32, device='cuda'),
                torch.full((b, max_ids), 9, dtype=torch.uint8, device='cuda'))

    def enqueue(rgb, ids, present):
        pflib.check(L.pf_panoptic_encode(seg.data_ptr(), 1, 1, b, h, w, rgb.data_ptr(), ids.data_ptr(), present.data_ptr(), pflib.stream_ptr()),
                    'pf_panoptic_encode')

    want = buffers()
    enqueue(*want)
    torch.cuda.synchronize()
    assert int(want[2].max()) == 1 and int(want[2].sum()) > 2
    out = buffers()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        enqueue(*out)
    for rep in range(REPLAYS):
        for t in out:
            t.fill_(1)          # `present` full of ones: only the call's own zero fill can clear it
        g.replay()
        torch.cuda.synchronize()
        for a, c in zip(out, want):
            assert torch.equal(a, c), rep
