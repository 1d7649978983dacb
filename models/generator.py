from gan_amd.generator import AIGGenerator  # noqa: F401
