"""Drop-in replacement of the reference's ``src.model`` package
(``src/model/{layers,models,loss}.py``): same class / function names,
constructor and forward signatures, return tuples and ``state_dict`` keys."""
from .layers import MLP, MHA, Encoder_Block, TransformerEncoder
from .models import Generator, Discriminator, simple_disc
from .loss import gradient_penalty, discriminator_loss, generator_loss

__all__ = ["MLP", "MHA", "Encoder_Block", "TransformerEncoder", "Generator", "Discriminator", "simple_disc",
           "gradient_penalty", "discriminator_loss", "generator_loss"]
