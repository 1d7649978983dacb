from gan_amd.discriminator import AIGDiscriminator  # noqa: F401
