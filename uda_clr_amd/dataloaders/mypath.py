"""``mypath.Path`` - the reference imports it (dataloaders/fundus_dataloader.py:6,19) but does not ship it."""
import os


class Path(object):
    @staticmethod
    def db_root_dir(database):
        if database == 'fundus':
            return os.environ.get('UDA_CLR_FUNDUS_ROOT', './Fundus/')
        raise NotImplementedError('Database %r not available.' % (database,))
