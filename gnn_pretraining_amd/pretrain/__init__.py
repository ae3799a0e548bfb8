"""Host-side mirror of the reference's src/pretrain package on libgnnmp."""
