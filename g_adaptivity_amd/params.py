"""The `opt` keys the hot path reads, with the reference's defaults.

The reference threads one plain dict `opt` through every constructor
(`src/params.py:199-303` argparse defaults, then `run_params` overrides
`src/params.py:106-134`).  Only the keys listed in SURVEY.md §5 ("Config /
flags") matter to the model; `hot_path_opt` returns exactly those so callers
(tests, bench, a reference-style training loop) can build models without the
reference's argparse front end.
"""
from __future__ import annotations


def hot_path_opt(**overrides) -> dict:
    opt = {
        # mesh / data
        'mesh_dims': [11, 11],              # params.py:37
        'data_type': 'randg',               # params.py:30
        'fix_boundary': True,               # params.py:67
        'eval_quad_points': 101,            # params.py:68
        # features
        'gnn_inc_feat_f': True,             # params.py:114
        'gnn_inc_feat_uu': True,            # params.py:115
        'gnn_inc_glob_feat_f': False,       # params.py:116
        'gnn_inc_glob_feat_uu': False,      # params.py:117
        'gnn_normalize': False,             # params.py:118
        'global_feat_dim': 8,               # params.py:133
        # model
        'conv_type': 'GRAND_plus',          # params.py:121
        'gat_plus_type': 'GAT_res_lap',     # params.py:122
        'enc': 'identity', 'dec': 'identity',
        'residual': True,                   # params.py:127
        'share_conv': True,                 # params.py:128
        'non_lin': 'identity',              # params.py:129
        'num_layers': 4,                    # params.py:130
        'time_step': 0.1,                   # params.py:131
        'hidden_dim': 8,                    # params.py:132
        'learn_step': False,                # params.py:267
        'self_loops': False,                # params.py:264
        'softmax_temp_type': None,          # params.py:265
        'softmax_temp': 2.0,                # params.py:266
        'reg_skew': False,                  # params.py:269
        'dropout': 0.0,                     # params.py:287
        # training
        'loss_type': 'mesh_loss',           # params.py:289 (run_params sets pde_loss; FEM tail is out of scope)
        'loss_fn': 'mse',
        'lr': 0.001, 'decay': 0.0,
        'device': 'cpu',
        # params.py:298 declares the string "True"; the pipeline's tf_sweep_args (params.py:172-177, run_pipeline.py:96-100)
        # turns it into the bool True before any model is built, and a bool makes GRAND_plusConv keep
        # stored_ei / stored_alpha (GRAND_plus.py:253).  Pass the string 'False' (or any non-bool) to skip that.
        'show_mesh_evol_plots': True,
        # not a reference key: False materialises the zero-padded encoder output, the full last-layer output and the padded
        # top gradient (the literal GNN.py:270,299 data flow) instead of their compact forms
        'compact_slots': True,
    }
    opt.update(overrides)
    return opt
