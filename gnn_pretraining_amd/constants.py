"""Dataset-level constants of the reference that the hot path needs
(src/data/data_setup.py:26,31-59; src/data/graph_properties.py GRAPH_PROPERTY_DIM)."""
PRETRAIN_TUDATASETS = ["MUTAG", "PROTEINS", "NCI1", "ENZYMES"]
DOMAIN_DIMENSIONS = {"MUTAG": 7, "PROTEINS": 4, "NCI1": 37, "ENZYMES": 21, "PTC_MR": 18, "Cora_NC": 1433,
                     "CiteSeer_NC": 3703, "Cora_LP": 1433, "CiteSeer_LP": 3703}
NUM_CLASSES = {"ENZYMES": 6, "PTC_MR": 2, "Cora_NC": 7, "CiteSeer_NC": 6, "Cora_LP": 2, "CiteSeer_LP": 2}
TASK_TYPES = {"ENZYMES": "graph_classification", "PTC_MR": "graph_classification", "Cora_NC": "node_classification",
              "CiteSeer_NC": "node_classification", "Cora_LP": "link_prediction", "CiteSeer_LP": "link_prediction"}
GRAPH_PROPERTY_DIM = 12
