from gan_amd.dataset import load_pt_dataset, pad_data, save_data_list_as_pt, unpad_data  # noqa: F401
