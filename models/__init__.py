"""Import paths of the reference (``from models.generator import AIGGenerator`` ...) served by gan_amd."""
