"""Hot-path configuration: the subset of the reference's ~170 flags (opts.py) that the accelerated path reads, with
the values of the BASELINE configs.  The reference's argparse/YAML layer itself is out of scope (SURVEY.md section 2
row 16); a Namespace produced by the reference's ``opts.parse_opts()`` works with ``gvl_amd.pdvc.build`` unchanged,
and ``make_opt`` builds an equivalent Namespace without it.
Values: opts.py defaults overlaid by cfgs/anet_tsp_ssvg.yml / cfgs/anet_c3d_ssvg.yml / cfgs/yc2_tsn_dvc.yml.
"""
import argparse

_COMMON = dict(
    hidden_dim=512, nheads=8, num_feature_levels=4, enc_n_points=4, dec_n_points=4, enc_layers=2, dec_layers=2,
    transformer_ff_dim=512, transformer_dropout_prob=0.1, with_box_refine=1, aux_loss=True, num_classes=1,
    max_eseq_length=10, share_caption_head=1, disable_mid_caption_heads=False,
    caption_decoder_type='standard', cap_nheads=1, cap_dec_n_points=4, cap_num_feature_levels=4, max_caption_len=30,
    att_hid_size=512, rnn_size=512, input_encoding_size=512, num_layers=1, drop_prob=0.5, clip_context_dim=512,
    wordRNN_input_feats_type='C', enable_pos_emb_for_captioner=False,
    set_cost_class=2, set_cost_bbox=0, set_cost_giou=4, set_cost_cl=2.0, cost_alpha=0.25, cost_gamma=2,
    set_cost_caption=0, caption_loss_coef=2, caption_loss_type='ce', caption_cost_type='loss',
    focal_alpha=0.25, focal_gamma=2.0, cls_loss_coef=2, count_loss_coef=0.5, bbox_loss_coef=0, giou_loss_coef=4,
    contrastive_loss_start_coef=0.0, lloss_gau_mask=1, lloss_beta=1,
    enable_contrastive=False, eval_disable_captioning=False, transformer_input_type='queries',
    lr=5e-5, weight_decay=1e-4, grad_clip=100.0, device='cuda',
)

CONFIGS = {
    # cfgs/anet_c3d_ssvg.yml  (BASELINE config 0: plumbing case)
    'anet_c3d_ssvg': dict(feature_dim=500, num_queries=30, frame_embedding_num=100, vocab_size=8517,
                          eval_batch_size=16, batch_size=1),
    # cfgs/anet_tsp_ssvg.yml  (BASELINE configs 1-3: the headline model)
    'anet_tsp_ssvg': dict(feature_dim=512, num_queries=30, frame_embedding_num=100, vocab_size=8517,
                          eval_batch_size=16, batch_size=1),
    'anet_tsp_msvg_dvc': dict(feature_dim=512, num_queries=30, frame_embedding_num=100, vocab_size=8517,
                              eval_batch_size=16, batch_size=1),
    # cfgs/yc2_tsn_dvc.yml    (BASELINE config 4: long videos, TSN features)
    'yc2_tsn_dvc': dict(feature_dim=3072, num_queries=100, frame_embedding_num=200, vocab_size=1607,
                        eval_batch_size=1, batch_size=1),
}


def make_opt(cfg='anet_tsp_ssvg', **overrides):
    if cfg not in CONFIGS:
        raise KeyError(f"unknown config {cfg!r}; have {sorted(CONFIGS)}")
    d = dict(_COMMON)
    d.update(CONFIGS[cfg])
    d['id'] = cfg
    unknown = set(overrides) - set(d)
    if unknown:
        raise KeyError(f"unknown option(s) {sorted(unknown)}")
    d.update(overrides)
    return argparse.Namespace(**d)
