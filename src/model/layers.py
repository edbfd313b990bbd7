from druggen_amd.model.layers import MLP, MHA, Encoder_Block, TransformerEncoder  # noqa: F401
