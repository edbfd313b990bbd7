from druggen_amd.model.loss import gradient_penalty, discriminator_loss, generator_loss  # noqa: F401
