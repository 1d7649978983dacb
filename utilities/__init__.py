"""Import paths of the reference (``from utilities.data_process import pad_data`` ...) served by gan_amd."""
