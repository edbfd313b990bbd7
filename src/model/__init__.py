"""Import-path shim: with this repository ahead of the reference on
``sys.path``, the reference's ``train.py`` / ``inference.py`` lines
``from src.model.models import Generator, Discriminator, simple_disc`` and
``from src.model.loss import discriminator_loss, generator_loss`` resolve to the
MI355X implementation unchanged."""
