def load_vposer(*args, **kwargs):
    raise RuntimeError('VPoser is not part of the LGD path.')
