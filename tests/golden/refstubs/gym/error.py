class DependencyNotInstalled(Exception):
    pass
