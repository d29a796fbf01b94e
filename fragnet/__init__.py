"""Import-path aliases: the names the reference's drivers import from ``fragnet.*`` resolve to the MI355X implementation in
``fragnet_amd`` (nothing of the reference is copied here), so that ``fragnet/train/finetune/finetune_gat2.py`` and
``fragnet/train/pretrain/pretrain_gat2.py`` run against this repository with their import lines unchanged:

    fragnet.model.gat.gat2            FragNet, FragNetFineTune           (finetune_gat2.py:121, pretrain_gat2.py:16)
    fragnet.model.gat.gat2_lite       FragNetFineTune                    (finetune_gat2.py:144)
    fragnet.model.gat.gat2_edge       FragNetFineTune                    (finetune_gat2.py:166)
    fragnet.model.gat.gat2_pretrain   FragNetPreTrain                    (finetune_gat2.py:216)
    fragnet.model.gat.pretrain_heads  FragNetPreTrain, PretrainTask      (pretrain_gat2.py:12)
    fragnet.dataset.data              collate_fn, collate_fn_pt          (finetune_gat2.py:6, pretrain_gat2.py:15)
    fragnet.dataset.dataset           load_pickle_dataset, load_data_parts (finetune_gat2.py:2, pretrain_gat2.py:6)
    fragnet.train.utils               EarlyStopping, TrainerFineTune     (finetune_gat2.py:4,9)
    fragnet.train.pretrain.pretrain_utils   Trainer                      (pretrain_gat2.py:13)

Model versions outside the accelerated hot path (masked pretraining heads, gcn / gat v1, DTA, CDRP; SURVEY.md section 2 rows
10-19) are named here only to fail with a clear message when constructed.
"""
