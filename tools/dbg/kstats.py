import csv, sys
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        if 'fl::' in r['Name']:
            print('%8.1f us x%-4s %s' % (float(r['AverageNs']) / 1e3, r['Calls'], r['Name'][:90]))
