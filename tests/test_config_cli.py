"""The reference's flag set and precedence rules (utils/config.py:34-95) for the export driver."""
import os

import yaml


def test_precedence_and_value_parsing(tmp_path):
    from panoptic_forecasting_amd import config as pfconfig
    from panoptic_forecasting_amd.export_bg import EXTRA_FLAGS
    ck = tmp_path / 'ck'
    ck.mkdir()
    (ck / 'config.yaml').write_text(yaml.dump({'task': 'bg', 'data': {'num_classes': 11, 'min_depth': 0.5},
                                               'model': {'final_w': 100}}))
    (ck / 'bg_model.pt').write_bytes(b'')
    cfg = tmp_path / 'val.yaml'
    cfg.write_text(yaml.dump({'data': {'min_depth': 0.1, 'gap_len': [3]}, 'training': {'batch_size': 2}}))
    p = pfconfig.load_config(EXTRA_FLAGS, ['--working_dir', str(tmp_path), '--config_file', str(cfg), '--load_model',
                                           str(ck / 'bg_model.pt'), '--no_convert', '--export_name', 'exp',
                                           '--extra_args', 'model.final_w', '2048', '--extra_args', 'data.cities', '[a, b]',
                                           '--extra_args', 'data.use_depths', 'true', '--extra_args', 'model.scale', '1.5'])
    assert p['task'] == 'bg' and p['data']['num_classes'] == 11            # from the checkpoint's config.yaml
    assert p['data']['min_depth'] == 0.1 and p['data']['gap_len'] == [3]   # --config_file wins over it
    assert p['model']['final_w'] == 2048 and p['model']['scale'] == 1.5    # --extra_args win over both
    assert p['data']['cities'] == ['a', 'b'] and p['data']['use_depths'] is True
    assert p['no_convert'] is True and p['export_name'] == 'exp' and p['convert_to_trainid'] is False
    assert p['seed'] == 1 and p['load_best_model'] is False and p['dist_url'] == 'env://'


def test_parse_value():
    from panoptic_forecasting_amd.config import parse_value
    assert parse_value('3') == 3 and parse_value('3.0') == 3.0 and parse_value('1e-3') == 1e-3
    assert parse_value('[1, 2.5, x, true]') == [1, 2.5, 'x', True]
    assert parse_value('data/cityscapes/') == 'data/cityscapes/'
