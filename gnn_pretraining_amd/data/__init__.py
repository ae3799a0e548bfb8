"""Datasets on disk and the loaders over them: the flat GraphStore format, the
processing logic of src/data/data_setup.py, and the pre-train / fine-tune loaders of
src/data/{pretrain,finetune}_data_loaders.py."""
