from druggen_amd.model.models import Generator, Discriminator, simple_disc  # noqa: F401
