"""Host-side mirror of src/finetune on libgnnmp (synthetic data)."""
