"""The reference's two entry points end to end on the GPU with synthetic stand-in data (src/pretrain/pretrain.py:284-349,
src/finetune/finetune.py:334-445): a short pre-training run writes the checkpoint (the reference's three keys + the random streams a resumed
run needs, loadable with the weights-only loader), and a Cora_NC-shaped fine-tune loads it, trains on the explicit-kernel engine, VALIDATES
THROUGH THE MODULE (whose encoder weight is then a strided view of the engine's padded slot -- round 3: this crashed) and returns test metrics."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_pretrain_then_node_classification_finetune_run_end_to_end(tmp_path, monkeypatch):
    from gnn_pretraining_amd.finetune import finetune as FT
    from gnn_pretraining_amd.models import finetune_model as FM
    from gnn_pretraining_amd.pretrain import pretrain as PT
    monkeypatch.setattr(PT, "OUTPUT_DIR", tmp_path / "pretrain")
    monkeypatch.setattr(FM, "PRETRAIN_OUTPUT_DIR", tmp_path / "pretrain")
    monkeypatch.setattr(FT, "OUTPUT_DIR", tmp_path / "finetune")
    path = PT.pretrain(PT.PretrainConfig(exp_name="s4", seed=7), epochs=1, steps_per_epoch=3, data_root=str(tmp_path / "data"), data_scale=0.03)
    ck = torch.load(path, weights_only=True)
    assert {"epoch", "model_state_dict", "val_metrics"} <= set(ck) and ck["rng_state"]["engine"]["step_count"] >= 3
    # load_pretrained_weights' default directory was bound at definition time: point it at the temporary one
    monkeypatch.setattr(FM.load_pretrained_weights, "__defaults__", (tmp_path / "pretrain",))
    test = FT.finetune(FT.FinetuneConfig("Cora_NC", "full_finetune", "s4", 7), epochs=3, data_root=str(tmp_path / "data"), data_scale=0.1)
    assert 0.0 <= test["test/accuracy"] <= 1.0 and "test/auc" in test
