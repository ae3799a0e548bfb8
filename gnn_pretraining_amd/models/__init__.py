"""Host-side mirror of the reference's src/models package (same class names, forward()
signatures and state_dict keys), executing on libgnnmp."""
from .gnn import GINBackbone, GINConv, GINLayer, InputEncoder, DROPOUT_RATE, GNN_HIDDEN_DIM, GNN_NUM_LAYERS  # noqa: F401
from .heads import DomainClassifierHead, GradientReversalLayer, MLPHead, MLPLinkPredictor  # noqa: F401
from .pretrain_model import PretrainableGNN  # noqa: F401
from .finetune_model import FinetuneGNN, create_finetune_model, load_pretrained_weights  # noqa: F401
