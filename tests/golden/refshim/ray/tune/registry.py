def register_env(name, creator):
    pass
